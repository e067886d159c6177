"""BatchNorm1d over the node axis with statistics reduced across the node partition, so that logits
equal the single-GPU ones (reference models/gcn.py:23,28 normalises over ALL N nodes). Same kernels as
rgb_experiment_amd.nn.BatchNorm1d; only the raw column sums (and the row count) are all-reduced."""
import torch.nn as nn

from ..nn.batchnorm import BatchNorm1d


class DistBatchNorm1d(BatchNorm1d):
    """Same parameters / buffers / state_dict keys as nn.BatchNorm1d."""

    def __init__(self, num_features, comm, **kw):
        super().__init__(num_features, **kw)
        self.comm = comm

    def _reduce(self, packed):
        return self.comm.all_reduce_sum_(packed)

    @classmethod
    def convert(cls, module, comm):
        """Replace every BatchNorm1d under `module` (in place), keeping parameters and buffers."""
        for name, child in list(module.named_children()):
            if isinstance(child, nn.BatchNorm1d) and not isinstance(child, cls):
                new = cls(child.num_features, comm, eps=child.eps, momentum=child.momentum)
                new.load_state_dict(child.state_dict())
                new.to(child.weight.device)
                new.train(child.training)
                setattr(module, name, new)
            else:
                cls.convert(child, comm)
        return module
