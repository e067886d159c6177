"""Data-parallel-over-nodes training of the reference's loop body (itexperiments.py:417-473) on a 1-D
node partition: one process per GPU, replicated parameters, per-propagate halo exchange, global
BatchNorm statistics, all-reduced loss and parameter gradients."""
import torch
import torch.nn.functional as F

from .comm import Comm
from .graph import install
from .nn import DistBatchNorm1d
from .plan import partition_bounds


class DistRunner:
    """model: any of the conv-stack models (GCN / GraphSAGE / GraphSAGE2 / APPNPStack), freshly built
    with the same seed on every rank. edge_index / x / y / masks: GLOBAL tensors (CPU or device); each
    rank keeps its node slice of x / y / masks and the whole edge list for index arithmetic."""

    def __init__(self, model, edge_index, x, y, masks, rank, world, device, lr=0.01, weight_decay=0.0,
                 comm=None, backend=None, exchange="auto", resident_features=True):
        self.comm = comm or Comm()
        self.rank, self.world, self.device = rank, world, device
        N = x.size(0)
        lo, hi = partition_bounds(N, world)[rank:rank + 2]
        self.lo, self.hi, self.N = lo, hi, N
        self.x = x[lo:hi].to(device).contiguous()
        self.y = y[lo:hi].to(device)
        self.masks = [m[lo:hi].to(device) for m in masks]
        self.edge_index = edge_index.to(device)
        # the conv layers see (x_local, token): the token's cache entries are the DistGraphs
        self.token = torch.zeros((2, 1), dtype=torch.int64, device=device)
        self.graphs = install(self.token, hi - lo, self.edge_index, N, self.comm, backend, exchange)
        if resident_features:
            for g in self.graphs.values():
                g.pin_resident(self.x)  # boundary rows of the static features are fetched once and kept
        self.model = DistBatchNorm1d.convert(model.to(device), self.comm)
        self.opt = torch.optim.Adam(self.model.parameters(), lr=lr, weight_decay=weight_decay)
        cnt = torch.tensor([float(m.sum()) for m in self.masks], dtype=torch.float64, device=device)
        self.mask_counts = self.comm.all_reduce_sum_(cnt).tolist()

    # ---- statistics used by bench.py --------------------------------------------------------
    def plan(self, loops_mode, kind):
        return self.graphs[loops_mode].plan(kind)

    def _sync_grads(self):
        grads = [p.grad for p in self.model.parameters() if p.grad is not None]
        flat = torch.cat([g.reshape(-1) for g in grads])
        self.comm.all_reduce_sum_(flat)
        off = 0
        for g in grads:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()

    def _nll_sum(self, res, m):
        """Sum over this rank's masked rows of NLLLoss(log_softmax(emb)): from the logits on the GPU."""
        if res["emb"].is_cuda:
            from .. import ops
            return ops.masked_ce_loss(res["emb"], self.y, m, reduction="sum")
        return F.nll_loss(res["out"][m], self.y[m], reduction="sum")

    def train_step(self, sync=True):
        """One training step. `sync=False`: returns this rank's share of the loss as a device tensor [1]
        (float64) instead of the all-reduced Python float — epoch() reduces everything once."""
        self.model.train()
        self.opt.zero_grad()
        res = self.model(self.x, self.token)
        m = self.masks[0]
        loss = self._nll_sum(res, m) / self.mask_counts[0]
        loss.backward()
        self._sync_grads()
        self.opt.step()
        part = loss.detach().double().reshape(1)
        if not sync:
            return part
        return self.comm.all_reduce_sum_(part.clone()).item()

    def evaluate(self, which, sync=True):
        """Eval forward + (masked NLL sum, correct count) of this rank's rows. `sync=True`: all-reduced and
        normalised Python floats (loss, accuracy, outputs); `sync=False`: the raw device tensor [2] and outputs."""
        self.model.eval()
        with torch.no_grad():
            res = self.model(self.x, self.token)
        m = self.masks[which]
        if res["emb"].is_cuda:
            from .. import ops
            stats = ops.masked_ce_accuracy(res["emb"], self.y, m)[[0, 2]]
        else:  # gloo/CPU tests
            out = res["out"]
            stats = torch.stack([F.nll_loss(out[m], self.y[m], reduction="sum"),
                                 (out[m].max(dim=1)[1] == self.y[m]).sum().float()]).double()
        if not sync:
            return stats, res
        stats = (self.comm.all_reduce_sum_(stats) / self.mask_counts[which]).tolist()
        return stats[0], stats[1], res

    def epoch(self):
        """1 train forward+backward+Adam, then val and test forwards, as the reference loop body. The five
        numbers the reference reads with .item() along the way are only used after the epoch: they are reduced
        over the ranks in ONE all-reduce and read back in ONE copy, so the queues drain once per epoch, not three
        times."""
        tl = self.train_step(sync=False)
        v, _ = self.evaluate(1, sync=False)
        s, _ = self.evaluate(2, sync=False)
        p = self.comm.all_reduce_sum_(torch.cat([tl, v, s])).tolist()
        cv, cs = self.mask_counts[1], self.mask_counts[2]
        return p[0], p[1] / cv, p[2] / cv, p[3] / cs, p[4] / cs

    def logits(self, training=False):
        self.model.train(training)
        with torch.no_grad():
            return self.model(self.x, self.token)["emb"]
