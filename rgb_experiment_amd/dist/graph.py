"""Distributed graph: per-rank CSRs (local-source and remote-source, forward and transposed) plus the
halo exchange, behind the same `propagate_*` entry points the conv layers already call."""
import torch

from .. import graph as _graph
from .comm import Comm
from .plan import PartitionPlan


class HipAggregator:
    """Product compute backend: rgbx_csr_build + rgbx_spmm_csr_f32 + rgbx_gather_rows_f32."""

    def prepare(self, agg, gather, n_rows, w):
        csr = _graph.build_csr(agg, gather, n_rows, _graph.LOOPS_KEEP)
        ws = None
        if w is not None:
            ws = w[csr.perm[:csr.nnz].long()].contiguous() if csr.nnz else torch.zeros(1, device=w.device)
        return csr, ws

    def run(self, handle, x, y=None, kind="dist_spmm"):
        from .. import ops
        csr, w = handle
        if y is None:
            return ops.spmm_raw(csr, w, None, x, kind=kind)
        return ops.spmm_raw(csr, w, None, x, y=y, a=1.0, b=1.0, out=y, kind=kind)

    def gather(self, x, idx):
        from .. import ops
        return ops.gather_rows(x, idx)


class _DistPropagate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dgraph, kind):
        ctx.dgraph, ctx.kind = dgraph, kind
        return dgraph._run(kind, "fwd", x)

    @staticmethod
    def backward(ctx, gy):
        return ctx.dgraph._run(ctx.kind, "bwd", gy.contiguous()), None, None


class DistGraph:
    """What `get_graph` returns on a rank of a partitioned run. `edge_index` is the GLOBAL int64 edge
    list (every rank holds it at set-up; only index arithmetic touches it), `num_nodes` the global N."""

    is_distributed = True

    def __init__(self, edge_index, num_nodes, loops_mode, comm=None, backend=None):
        self.edge_index, self.N_global, self.loops_mode = edge_index, int(num_nodes), loops_mode
        self.comm = comm or Comm()
        self.backend = backend or HipAggregator()
        self._kinds = {}

    def _get(self, kind):
        st = self._kinds.get(kind)
        if st is None:
            plan = PartitionPlan(self.edge_index, self.N_global, self.comm.world, self.comm.rank,
                                 self.loops_mode, kind)
            st = {"plan": plan}
            for name, half in (("fwd", plan.fwd), ("bwd", plan.bwd)):
                st[name] = {
                    "half": half,
                    "loc": self.backend.prepare(half.loc_agg, half.loc_gather, half.n_local, half.loc_w),
                    "rem": self.backend.prepare(half.rem_agg, half.rem_gather, half.n_local, half.rem_w)
                    if half.n_halo else None,
                }
            self._kinds[kind] = st
        return st

    def plan(self, kind):
        return self._get(kind)["plan"]

    def _run(self, kind, direction, x):
        d = self._get(kind)[direction]
        half = d["half"]
        work = recv = None
        if self.comm.world > 1:
            send = self.backend.gather(x, half.send_idx) if half.n_send else x.new_empty((0, x.size(1)))
            recv, work = self.comm.all_to_all_rows(send, half.send_counts, half.recv_counts)
        out = self.backend.run(d["loc"], x, kind=f"dist_{direction}_local")  # overlaps the exchange
        if work is not None:
            work.wait()
            if d["rem"] is not None:
                out = self.backend.run(d["rem"], recv, y=out, kind=f"dist_{direction}_remote")
        return out

    def propagate(self, x, kind):
        return _DistPropagate.apply(x, self, kind)


def install(token_edge_index, n_local, edge_index, num_nodes, comm=None, backend=None):
    """Register DistGraphs so that conv layers called with (x_local, token_edge_index) aggregate over
    the partitioned global graph. Returns {loops_mode: DistGraph}."""
    graphs = {}
    for mode in (_graph.LOOPS_KEEP, _graph.LOOPS_ADD_REMAINING, _graph.LOOPS_REMOVE_ADD):
        g = DistGraph(edge_index, num_nodes, mode, comm, backend)
        _graph.register_graph(token_edge_index, n_local, mode, g)
        graphs[mode] = g
    return graphs
