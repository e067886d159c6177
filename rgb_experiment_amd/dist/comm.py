"""Thin collective layer over torch.distributed. Product: backend 'nccl' (= RCCL over xGMI), device
tensors, the all-to-all runs asynchronously on RCCL's stream so the local-edge SpMM overlaps it.
Rehearsal/tests: backend 'gloo'; device tensors are staged through host memory (gloo has no device
all-to-all), CPU tensors go straight through."""
import torch
import torch.distributed as dist


class _Done:
    def wait(self):
        return True


class Comm:
    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.backend = dist.get_backend(group) if dist.is_initialized() else "none"

    def all_to_all_rows(self, send, send_counts, recv_counts):
        """Row-wise all-to-all: rank q receives send[offs[q]:offs[q+1]] of every peer, concatenated in
        rank order. Returns (recv, work); call work.wait() before reading recv."""
        n_recv = int(sum(recv_counts))
        recv = torch.empty((n_recv, send.size(1)), dtype=send.dtype, device=send.device)
        if self.world == 1:
            return recv, _Done()
        if self.backend == "nccl" or not send.is_cuda:
            work = dist.all_to_all_single(recv, send.contiguous(), list(recv_counts), list(send_counts),
                                          group=self.group, async_op=True)
            return recv, work
        host_recv = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_to_all_single(host_recv, send.cpu().contiguous(), list(recv_counts), list(send_counts),
                               group=self.group)
        recv.copy_(host_recv)
        return recv, _Done()

    def all_reduce_sum_(self, t):
        if self.world == 1:
            return t
        if self.backend == "nccl" or not t.is_cuda:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            return t
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
        t.copy_(h)
        return t

    def barrier(self):
        if self.world > 1:
            dist.barrier(group=self.group)
