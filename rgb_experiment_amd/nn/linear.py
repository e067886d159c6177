"""nn.Linear whose backward takes dW = dYᵀX (and db) from rgbx_gemm_tn_f32 (csrc/gemm.hip).

The dense layers around the aggregation (DAGNN's and GIN's MLPs, reference models/dagnn.py:66-67, gin.py:14-40) are
`torch.nn.Linear` in the reference; under autograd their weight gradient is a tall-skinny [out, N] x [N, in] product
that hipBLASLt runs at 2.4-2.7 ms for N = 2 M, 128 x 128 (profiles/old/r02_L_dagnn_kernel_stats.csv) against 0.66 ms for
the split-K MFMA kernel of this library. Same parameters, same state_dict keys, same forward arithmetic."""
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


class Linear(nn.Linear):
    def forward(self, x):
        if x.is_cuda and x.dim() == 2:
            return ops.linear(x, self.weight, self.bias)
        return F.linear(x, self.weight, self.bias)
