"""Distributed graph: per-rank CSRs (local-source and remote-source, forward and transposed) plus the
halo exchange, behind the same `propagate_*` entry points the conv layers already call."""
import torch

from .. import graph as _graph
from .comm import Comm
from .plan import PartitionPlan, edge_weights, partition_bounds, rewrite_global


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

    def run_rows(self, handle, x, lo, hi, out, kind="dist_spmm"):
        """Rows [lo, hi) of the same product, written into `out` ([hi - lo, d]); False when the CSR carries a
        hub-row plan (row ids in the plan are absolute: the caller then takes the unchunked route)."""
        from .. import ops
        csr, w = handle
        if csr.split is not None:
            return False
        if hi > lo:
            view = _graph.CSR(csr.rowptr[lo:hi + 1], csr.col, csr.perm, hi - lo, csr.nnz, None)
            ops.spmm_raw(view, w, None, x, out=out, kind=kind)
        return True

    def gather(self, x, idx):
        from .. import ops
        return ops.gather_rows(x, idx)

    def appnp(self, handle, h, K, alpha, kind="dist_appnp_colshard"):
        from .. import ops
        csr, w = handle
        return ops.appnp_raw(csr, w, h, K, alpha, kind=kind)

    def scatter_add(self, src, idx, dst):
        from .. import ops
        return ops.scatter_add_rows(src, idx, dst)

    def prepare_rect(self, agg, gather, n_tgt, n_src):
        """Rectangular graph: targets [0, n_tgt) aggregate from sources [0, n_src) (local rows + halo)."""
        return _RectGraph(_graph.build_csr(agg, gather, n_tgt, _graph.LOOPS_KEEP),
                          _graph.build_csr(gather, agg, n_src, _graph.LOOPS_KEEP))

    def gat(self, rect, x_ext, att_src, att_dst, n_tgt, H, C, slope):
        from .. import ops
        return ops._GATAttend.apply(x_ext, att_src, att_dst, rect, H, C, slope)


class _ExtGraph:
    """What ops._PropagateLinear reads of a graph, for a rectangular [targets x (local; halo)] CSR."""

    def __init__(self, fwd, w):
        self.fwd, self.w, self.inv_deg = fwd, w, None


class _RectGraph:
    def __init__(self, fwd, bwd):
        self.fwd, self.bwd = fwd, bwd


class _HaloGather(torch.autograd.Function):
    """x_local [n_local, d] -> [x_local; halo rows] with the boundary rows fetched by one all-to-all.
    Backward sends the halo part of the gradient home and adds it to the owners' rows (one
    unique-index scatter-add per peer, in rank order: deterministic)."""

    @staticmethod
    def forward(ctx, x, dgraph, half):
        ctx.dgraph, ctx.half = dgraph, half
        x = x.contiguous()
        send = dgraph.backend.gather(x, half.send_idx) if half.n_send else x.new_empty((0, x.size(1)))
        recv, work = dgraph.comm.all_to_all_rows(send, half.send_counts, half.recv_counts)
        work.wait()
        return torch.cat([x, recv], dim=0)

    @staticmethod
    def backward(ctx, g_ext):
        dgraph, half = ctx.dgraph, ctx.half
        n = half.n_local
        g_loc = g_ext[:n].clone()
        back, work = dgraph.comm.all_to_all_rows(g_ext[n:].contiguous(), half.recv_counts, half.send_counts)
        work.wait()
        off = 0
        for cnt in half.send_counts:
            if cnt:
                dgraph.backend.scatter_add(back[off:off + cnt], half.send_idx[off:off + cnt], g_loc)
            off += cnt
        return g_loc, None, None


class _DistPropagate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dgraph, kind):
        ctx.dgraph, ctx.kind = dgraph, kind
        return dgraph._run(kind, "fwd", x)

    @staticmethod
    def backward(ctx, gy):
        return ctx.dgraph._run(ctx.kind, "bwd", gy.contiguous()), None, None


class _DistAPPNPColumns(torch.autograd.Function):
    """APPNP under the reshard scheme: the recurrence z <- (1-alpha) A_hat z + alpha h acts on every column
    independently, so ONE transpose to column shards, all K propagates on the whole graph at width d/P, and
    ONE transpose back replace 2K all-to-alls. Backward: the same recurrence on the transposed graph."""

    @staticmethod
    def forward(ctx, h, dgraph, K, alpha):
        ctx.dgraph, ctx.K, ctx.alpha = dgraph, K, alpha
        return dgraph._appnp_columns("fwd", h.contiguous(), K, alpha)

    @staticmethod
    def backward(ctx, gy):
        return ctx.dgraph._appnp_columns("bwd", gy.contiguous(), ctx.K, ctx.alpha), None, None, None


class DistGraph:
    """What `get_graph` returns on a rank of a partitioned run. `edge_index` is the GLOBAL int64 edge
    list (every rank holds it at set-up; only index arithmetic touches it), `num_nodes` the global N.

    Two exchange schemes, chosen per (kind, width) by the bytes a rank must receive per propagate:
      * "halo"    — ship the boundary rows a rank's edges gather from (n_halo * d * 4 B), overlap with
                    the local-edge SpMM. Wins when the partition has a small boundary.
      * "reshard" — all-to-all transpose the row-sharded [n_local, d] activations into column shards
                    [N, d/P], run the WHOLE graph's SpMM on d/P columns with no halo, transpose back
                    (2 * n_local * d * 4 * (P-1)/P B). Wins on expander-like graphs (the benchmark's
                    uniform random graph: every remote row is a boundary row).
    """

    is_distributed = True

    def __init__(self, edge_index, num_nodes, loops_mode, comm=None, backend=None, exchange="auto"):
        self.edge_index, self.N_global, self.loops_mode = edge_index, int(num_nodes), loops_mode
        self.comm = comm or Comm()
        self.backend = backend or HipAggregator()
        self.exchange = exchange
        self._kinds, self._full, self._choice = {}, {}, {}
        self._resident = None
        b = partition_bounds(self.N_global, self.comm.world)
        self.bounds = b
        self.n_local = b[self.comm.rank + 1] - b[self.comm.rank]
        self.row_counts = [b[q + 1] - b[q] for q in range(self.comm.world)]

    # ---- halo scheme ---------------------------------------------------------------------------
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

    # ---- resident input features ---------------------------------------------------------------
    # The first conv of every model aggregates the STATIC feature matrix. Its boundary rows never change, so
    # they are fetched once and kept in HBM next to the local rows ([local; halo], at most N*d*4 B per rank:
    # 1 GB on the 2M-node benchmark, of 288 GB); every later propagate of that tensor is one SpMM over the
    # combined CSR with no exchange at all. Nothing computed is cached: the aggregation runs every time.
    def pin_resident(self, x):
        """Declare `x` (this rank's rows of the input features, never written again) resident."""
        self._resident = {"ptr": x.data_ptr(), "shape": tuple(x.shape), "version": x._version, "ext": {}}

    def is_resident(self, x):
        r = self._resident
        return (r is not None and x.data_ptr() == r["ptr"] and tuple(x.shape) == r["shape"]
                and x._version == r["version"] and not x.requires_grad)

    def _resident_ext(self, x, half, key):
        """[x; halo rows of x] for the plan `half` belongs to, exchanged on first use (collective: every rank
        reaches this with its own resident tensor at the same point of the model)."""
        ext = self._resident["ext"].get(key)
        if ext is None:
            ext = x
            if self.comm.world > 1:
                send = self.backend.gather(x, half.send_idx) if half.n_send else x.new_empty((0, x.size(1)))
                recv, work = self.comm.all_to_all_rows(send, half.send_counts, half.recv_counts)
                work.wait()
                if half.n_halo:
                    ext = torch.cat([x, recv], dim=0)
            self._resident["ext"][key] = ext
        return ext

    def _ext_csr(self, kind):
        """CSR of this rank's targets over [local; halo] sources (+ per-edge weights in its slot order)."""
        st = self._get(kind)
        half = st["fwd"]["half"]
        if "ext_csr" not in st:
            agg = torch.cat([half.loc_agg, half.rem_agg])
            gather = torch.cat([half.loc_gather, half.n_local + half.rem_gather])
            w = None if half.loc_w is None else torch.cat([half.loc_w, half.rem_w])
            st["ext_csr"] = self.backend.prepare(agg, gather, half.n_local, w)
        return st["ext_csr"], half

    def _run_resident(self, kind, x):
        handle, half = self._ext_csr(kind)
        return self.backend.run(handle, self._resident_ext(x, half, kind), kind="dist_fwd_resident")

    def fused_resident_ok(self, x, in_channels, out_channels, root):
        """The fused aggregate+transform kernel can serve this propagate: resident input features on the GPU
        through the HIP backend, supported widths."""
        from .. import _lib
        return (x.is_cuda and isinstance(self.backend, HipAggregator) and self.is_resident(x)
                and bool(_lib.load().rgbx_spmm_linear_supported(in_channels, out_channels, int(root))))

    def propagate_linear(self, x, kind, weight, bias, root_weight):
        """(P x) W^T + b (+ x Wr^T) for the resident input features: rgbx_spmm_linear_f32 over the [local; halo]
        CSR, no exchange; every weighting kind arrives as per-edge weights (plan.edge_weights)."""
        from .. import ops
        (csr, ws), half = self._ext_csr(kind)
        x_ext = self._resident_ext(x, half, kind)
        need_z = torch.is_grad_enabled() and weight.requires_grad
        return ops._PropagateLinear.apply(x_ext, _ExtGraph(csr, ws), "gcn", weight, bias, need_z, root_weight, x)

    def _run_halo(self, kind, direction, x):
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

    # ---- reshard scheme ------------------------------------------------------------------------
    def _get_full(self, kind):
        st = self._full.get(kind)
        if st is None:
            src, dst = rewrite_global(self.edge_index, self.N_global, self.loops_mode)
            w = edge_weights(src, dst, self.N_global, kind)
            st = {"nnz": int(src.numel()),
                  "fwd": self.backend.prepare(dst, src, self.N_global, w),
                  "bwd": self.backend.prepare(src, dst, self.N_global, w)}
            self._full[kind] = st
        return st

    def _to_columns(self, x):
        """[n_local, d] row shard -> [N, d/P] column shard (peer q receives my rows of its column slice)."""
        P, n_loc, dc = self.comm.world, self.n_local, x.size(1) // self.comm.world
        send = x.view(n_loc, P, dc).permute(1, 0, 2).reshape(P * n_loc, dc)
        cols, work = self.comm.all_to_all_rows(send, [n_loc] * P, self.row_counts)
        work.wait()
        return cols

    def _to_rows(self, y):
        """[N, d/P] column shard -> [n_local, d] row shard."""
        P, n_loc, dc = self.comm.world, self.n_local, y.size(1)
        back, work = self.comm.all_to_all_rows(y, self.row_counts, [n_loc] * P)
        work.wait()
        return back.view(P, n_loc, dc).permute(1, 0, 2).reshape(n_loc, P * dc)

    # pieces the outgoing transpose is cut into (1 = one all-to-all after the whole SpMM); every piece costs one SpMM
    # launch per peer, so fewer pieces at larger world sizes keep the host ahead of the GPU
    reshard_chunks = None

    def _chunks(self):
        if self.reshard_chunks is not None:
            return int(self.reshard_chunks)
        return 4 if self.comm.world <= 4 else 2

    def _run_reshard(self, kind, direction, x):
        handle = self._get_full(kind)[direction]
        cols = self._to_columns(x)
        tag = f"dist_{direction}_colshard"
        out = self._reshard_pipelined(handle, cols, tag)
        if out is not None:
            return out
        return self._to_rows(self.backend.run(handle, cols, kind=tag))

    def _reshard_pipelined(self, handle, cols, tag):
        """SpMM and the transpose back to row shards, overlapped: every peer's block of destination rows is cut
        into `reshard_chunks` pieces; piece c of ALL peers is aggregated (one launch per peer block, straight into
        the send buffer) and handed to an asynchronous all-to-all, which runs on RCCL's stream while piece c + 1
        is being aggregated. Only the last piece's exchange is exposed. Returns None when the backend cannot
        aggregate row ranges (hub-row plan, test doubles without run_rows)."""
        C = self._chunks()
        run_rows = getattr(self.backend, "run_rows", None)
        if C <= 1 or run_rows is None:
            return None
        P, n_loc, dc, b = self.comm.world, self.n_local, cols.size(1), self.bounds
        cut = lambda n, c: (n * c) // C  # piece c of a block of n rows = rows [cut(n, c), cut(n, c + 1))
        pending = []
        for c in range(C):
            counts = [cut(b[q + 1] - b[q], c + 1) - cut(b[q + 1] - b[q], c) for q in range(P)]
            send = cols.new_empty((sum(counts), dc))
            off = 0
            for q in range(P):
                lo = b[q] + cut(b[q + 1] - b[q], c)
                if run_rows(handle, cols, lo, lo + counts[q], send[off:off + counts[q]], kind=tag) is False:
                    if pending:
                        raise RuntimeError("reshard: backend refused a row range after accepting one")
                    return None
                off += counts[q]
            m = cut(n_loc, c + 1) - cut(n_loc, c)
            recv, work = self.comm.all_to_all_rows(send, counts, [m] * P)
            pending.append((recv, work, m, send))  # `send` stays referenced until its exchange has been waited on
        parts = []
        for recv, work, m, _send in pending:
            work.wait()
            parts.append(recv.view(P, m, dc).permute(1, 0, 2).reshape(m, P * dc))
        return torch.cat(parts, dim=0)

    def _appnp_columns(self, direction, h, K, alpha):
        out = self.backend.appnp(self._get_full("gcn")[direction], self._to_columns(h), K, alpha,
                                 kind=f"dist_{direction}_appnp_colshard")
        return self._to_rows(out)

    def appnp(self, h, K, alpha):
        """K-step APPNP on the partitioned graph; `None` when the per-iteration path should be used."""
        if self.comm.world > 1 and self.scheme(h.size(1)) == "reshard":
            return _DistAPPNPColumns.apply(h, self, K, alpha)
        return None

    # ---- choice --------------------------------------------------------------------------------
    def halo_rows(self):
        """Rows this rank would receive per forward propagate under the halo scheme."""
        if "halo_rows" not in self._choice:
            src, dst = rewrite_global(self.edge_index, self.N_global, self.loops_mode)
            lo, hi = self.bounds[self.comm.rank], self.bounds[self.comm.rank + 1]
            remote = (dst >= lo) & (dst < hi) & ((src < lo) | (src >= hi))
            self._choice["halo_rows"] = int(torch.unique(src[remote]).numel())
        return self._choice["halo_rows"]

    def scheme(self, d):
        """'halo' or 'reshard' for feature width d (all ranks reach the same answer). Reshard needs
        d divisible by the world size."""
        P = self.comm.world
        if P == 1 or d % P != 0 or self.exchange == "halo":
            return "halo"
        if self.exchange == "reshard":
            return "reshard"
        key = ("scheme", d)
        if key not in self._choice:
            halo_bytes = torch.tensor([float(self.halo_rows()) * d * 4], dtype=torch.float64,
                                      device=self.edge_index.device)
            halo_bytes = self.comm.all_reduce_sum_(halo_bytes).item() / P  # mean over ranks
            reshard_bytes = 2.0 * (self.N_global / P) * d * 4 * (P - 1) / P
            self._choice[key] = "reshard" if reshard_bytes < halo_bytes else "halo"
        return self._choice[key]

    def _run(self, kind, direction, x):
        if direction == "fwd" and self.is_resident(x):
            return self._run_resident(kind, x)
        if self.comm.world > 1 and self.scheme(x.size(1)) == "reshard":
            return self._run_reshard(kind, direction, x.contiguous())
        return self._run_halo(kind, direction, x)

    def propagate(self, x, kind):
        return _DistPropagate.apply(x, self, kind)

    # ---- GAT: attention needs every in-edge of a target in one softmax, so the halo rows are appended to
    # the local rows and ONE rectangular CSR (local targets x [local; halo] sources) feeds the same fused
    # kernels as the single-GPU path. a_src of halo rows is recomputed locally from the received rows.
    def gat(self, h, att_src, att_dst, H, C, slope, weight=None):
        """Attention aggregate of h = x W^T. With `weight` given, `h` is the resident input x itself and the
        transform is applied here, to local and halo rows alike."""
        st = self._kinds.get("gat")
        if st is None:
            plan = PartitionPlan(self.edge_index, self.N_global, self.comm.world, self.comm.rank,
                                 self.loops_mode, "sum")
            f = plan.fwd
            agg = torch.cat([f.loc_agg, f.rem_agg])
            gather = torch.cat([f.loc_gather, f.n_local + f.rem_gather])
            st = {"plan": plan, "rect": self.backend.prepare_rect(agg, gather, f.n_local, f.n_local + f.n_halo)}
            self._kinds["gat"] = st
        half = st["plan"].fwd
        if weight is not None:
            # resident input features: h of the halo rows is recomputed from the resident copies (a GEMM over
            # n_local + n_halo rows) instead of being exchanged, forward and backward
            from .. import ops
            x_ext = ops.linear(self._resident_ext(h, half, "gat"), weight)
        else:
            x_ext = _HaloGather.apply(h, self, half) if self.comm.world > 1 else h
        return self.backend.gat(st["rect"], x_ext, att_src, att_dst, half.n_local, H, C, slope)


def install(token_edge_index, n_local, edge_index, num_nodes, comm=None, backend=None, exchange="auto"):
    """Register DistGraphs so that conv layers called with (x_local, token_edge_index) aggregate over
    the partitioned global graph. Returns {loops_mode: DistGraph}."""
    graphs = {}
    for mode in (_graph.LOOPS_KEEP, _graph.LOOPS_ADD_REMAINING, _graph.LOOPS_REMOVE_ADD):
        g = DistGraph(edge_index, num_nodes, mode, comm, backend, exchange)
        _graph.register_graph(token_edge_index, n_local, mode, g)
        graphs[mode] = g
    return graphs
