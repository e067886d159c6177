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
        self.replicated_rows = None  # row count of inputs EVERY rank holds in full (dist.ReplicaGraph): no reduction

    def _reduce(self, packed):
        return self.comm.all_reduce_sum_(packed)

    def _reducer(self, x):
        """An input with `replicated_rows` rows is the same matrix on every rank (the replicated first layer of
        dist.ReplicaGraph): its local column sums ARE the global ones, and in backward every rank normalises its own
        partial gradient — BatchNorm's backward is linear in the incoming gradient, so the partial results add up to
        the true gradient in the parameter all-reduce."""
        if self.replicated_rows is not None and x.size(0) == self.replicated_rows:
            return lambda packed: packed
        return self._reduce

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
