"""One epoch of the reference loop body (itexperiments.py:417-473) captured as a HIP graph.

On small graphs (Cora-sized: 2.7 k nodes) an epoch is ~100 kernel launches of a few microseconds
each, so the loop is bound by launch latency and by the host round-trips of ``loss.item()`` and the
accuracy computations, not by the kernels. Here the whole epoch — train forward, masked NLL, backward,
Adam step, then the val and test eval forwards with their masked NLL / arg-max accuracy — is captured
once into a hipGraph (``torch.cuda.CUDAGraph``; every rgbx_* entry point launches asynchronously on the
capturing stream, so the C ABI calls are captured like any other kernel) and replayed per epoch; the
host reads nine numbers per epoch in one copy.

The arithmetic is the eager path's, kernel for kernel (same launches, same order), so losses and
parameters are bitwise identical to the eager loop.
"""
import copy

import torch

from . import ops


class GraphedEpoch:
    """capture(): build the graph; run(): replay it and return
    (train_loss, train_acc, val_loss, val_acc, test_loss, test_acc)."""

    def __init__(self, net, optimizer, fwd, y, masks, share_eval_forward=False):
        self.net, self.opt, self.fwd, self.y = net, optimizer, fwd, y
        self.share_eval_forward = share_eval_forward
        self.train_mask, self.val_mask, self.test_mask = masks
        self.graph = None
        self.stats = None       # [3, 3] float64: rows train / val / test, columns (nll sum, count, correct)
        self.train_out = None   # static training loss of the captured epoch
        self.eval_out = None    # (kept for compatibility: the eval forwards no longer materialise outputs)

    def _epoch_body(self):
        net, y = self.net, self.y
        from .models._stack import masked_ce
        net.train()
        self.opt.zero_grad(set_to_none=True)
        # NLLLoss(log_softmax(emb)[mask], y[mask]) (reference :429) and the train accuracy (:434) of the training
        # forward; log-softmax is never written, and where the model's last conv can take the loss into its kernel
        # (models/_stack.masked_ce) neither are the logits
        loss, train_stats = masked_ce(net, self.fwd, y, self.train_mask)
        loss.backward()
        self.opt.step()
        net.eval()
        with torch.no_grad():
            if self.share_eval_forward:  # one eval forward, two masks: both statistics sets from the last conv's kernel
                from .models._stack import masked_ce_pair  # where it takes the loss (else from the one set of logits)
                val_stats, test_stats = masked_ce_pair(net, self.fwd, y, self.val_mask, self.test_mask).unbind(0)
            else:  # the reference's two identical eval forwards (itexperiments.py:464,470)
                val_stats = masked_ce(net, self.fwd, y, self.val_mask)[1]
                test_stats = masked_ce(net, self.fwd, y, self.test_mask)[1]
        return loss, None, torch.stack([train_stats, val_stats, test_stats])

    def capture(self, warmup=3):
        """Warm up on a side stream (allocator + lazily built CSRs), restore every piece of state the
        warm-up touched, then capture. Nothing of the warm-up survives: epoch 0 of the replayed graph
        starts from the same parameters, BatchNorm statistics and Adam state as the eager loop would."""
        if self.opt.state:
            raise RuntimeError("GraphedEpoch.capture expects a fresh optimizer (no steps taken yet)")
        net_state = copy.deepcopy(self.net.state_dict())
        rng = torch.cuda.get_rng_state()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        try:
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    self._epoch_body()
        finally:  # whatever happens, the caller gets its initial parameters / BatchNorm statistics back
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.net.load_state_dict(net_state)
        # Adam's state tensors (step, exp_avg, exp_avg_sq) must EXIST before capture — created inside the
        # capture they would be re-zeroed by every replay — and must hold their fresh values: zero them
        # in place instead of dropping them.
        for st in self.opt.state.values():
            for v in st.values():
                if torch.is_tensor(v):
                    v.zero_()
        torch.cuda.set_rng_state(rng)
        self.opt.zero_grad(set_to_none=True)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self.train_out, self.eval_res, self.stats = self._epoch_body()
        self.graph = graph
        return self

    def run(self):
        self.graph.replay()
        # a replay updates the parameters without moving their Python-side version counters and without running
        # optimizer hooks: retire what is cached per parameter state (ops.weight_t, _eval_operands)
        ops.note_weights_changed()
        s = self.stats.tolist()  # the one host sync of the epoch
        out = []
        for nll, cnt, correct in s:
            out += [nll / cnt, correct / cnt] if cnt else [float("nan"), float("nan")]
        return tuple(out)
