"""Distributed graph: per-rank CSRs (local-source and remote-source, forward and transposed) plus the
halo exchange, behind the same `propagate_*` entry points the conv layers already call."""
import torch

from .. import graph as _graph
from .comm import Comm
from .plan import GridPlan, PartitionPlan, partition_bounds, rewrite_global, subsets_from_slices


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

    def run_rows(self, handle, x, lo, hi, out, kind="dist_spmm", accumulate=False):
        """Rows [lo, hi) of the same product, written into `out` ([hi - lo, d]) — or ADDED to it (`accumulate`: the
        second and later source pieces of a propagate whose sources arrive piece by piece); False when the CSR carries
        a hub-row plan (row ids in the plan are absolute: the caller then takes the unchunked route)."""
        from .. import ops
        csr, w = handle
        if csr.split is not None:
            return False
        if hi > lo:
            view = _graph.CSR(csr.rowptr[lo:hi + 1], csr.col, csr.perm, hi - lo, csr.nnz, None)
            if accumulate:
                ops.spmm_raw(view, w, None, x, y=out, a=1.0, b=1.0, out=out, kind=kind)
            else:
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
        want_grad = torch.is_grad_enabled() and (x_ext.requires_grad or att_src.requires_grad or att_dst.requires_grad)
        return ops._GATAttend.apply(x_ext, att_src, att_dst, rect, H, C, slope, None, None, want_grad)


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
        recv, work = dgraph.comm.all_to_all_rows(send, half.send_counts, half.recv_counts, tag="halo")
        work.wait()
        return torch.cat([x, recv], dim=0)

    @staticmethod
    def backward(ctx, g_ext):
        dgraph, half = ctx.dgraph, ctx.half
        n = half.n_local
        g_loc = g_ext[:n].clone()
        back, work = dgraph.comm.all_to_all_rows(g_ext[n:].contiguous(), half.recv_counts, half.send_counts,
                                                 tag="halo")
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
    list, needed only while the per-rank structures are being built (index arithmetic); `release_edges()` drops
    it afterwards, so what stays resident is this rank's share. `num_nodes` is the global N.

    Exchange schemes, chosen per feature width by a cost model (`scheme`):
      * "halo"     - ship the boundary rows a rank's edges gather from (n_halo * d * 4 B), overlapped with
                     the local-edge SpMM. Wins when the partition has a small boundary.
      * "reshard"  - all-to-all transpose the row-sharded [n_local, d] activations into column shards
                     [N, d/P], aggregate the WHOLE graph on d/P columns, transpose back
                     (2 * n_local * d * 4 * (P-1)/P B). Least bytes on expander-like graphs (the benchmark's
                     uniform random graph: every remote row is a boundary row).
      * "gridRxC"  - R row groups x C column slices (plan.GridHalf; reshard is R = 1): the rank aggregates the
                     rows of its row group at width d/C. At P = 8, d = 128 the 2 x 4 grid keeps d/C = 32 floats
                     = one full 128-byte line per gathered row (a 16-float row costs the same line), i.e. half the
                     line requests of the plain transpose, for twice its inbound bytes.
    """

    is_distributed = True
    # APPNP under the transpose: the last of the K steps returns in pieces behind their aggregation (_appnp_columns);
    # False = all K steps in one call and one outbound exchange (round 2's form)
    appnp_return_in_pieces = True
    _appnp_return_pieces = 1

    def __init__(self, edge_index, num_nodes, loops_mode, comm=None, backend=None, exchange="auto", pieces=None,
                 plan_from_slices=None):
        self.edge_index, self.N_global, self.loops_mode = edge_index, int(num_nodes), loops_mode
        self.comm = comm or Comm()
        # OPT-IN (RGBX_PLAN_FROM_SLICES=1): the halo and grid plans from this rank's 1/P of the edge list + one all-to-all
        # of edge records per direction (plan.subsets_from_slices), instead of every rank scanning the whole list. The same
        # plans bit for bit (tests/test_dist_gloo.py); a COLLECTIVE at the first plan of a group size — every rank builds its
        # plans at the same points of the same model code, as it reaches the exchanges themselves. Off by default: every
        # rank holds the whole edge list anyway (experiment()'s interface hands it over), the 1 x P transpose needs all of
        # it, and no multi-GPU hardware has been there to show the collective pays for the scan it saves.
        import os
        if plan_from_slices is None:
            plan_from_slices = os.environ.get("RGBX_PLAN_FROM_SLICES", "0") == "1"
        self.plan_from_slices = bool(plan_from_slices)
        self._subsets = {}
        self.backend = backend or HipAggregator()
        self.exchange = exchange
        self._kinds, self._grid, self._choice = {}, {}, {}
        self._resident = None
        b = partition_bounds(self.N_global, self.comm.world)
        self.bounds = b
        self.n_local = b[self.comm.rank + 1] - b[self.comm.rank]
        self.row_counts = [b[q + 1] - b[q] for q in range(self.comm.world)]
        # pieces the outgoing exchange of the grid schemes is cut into: piece k's all-to-all is in flight while
        # piece k + 1 is aggregated (one SpMM launch per piece), so 1/pieces of it is exposed
        # (given: fixed; else chosen per width by the cost model from the measured link rate and all-to-all latency)
        self.pieces = int(pieces) if pieces else 4
        self._pieces_fixed = bool(pieces)

    def _edges(self):
        if self.edge_index is None:
            raise RuntimeError("DistGraph: the global edge list was released (release_edges()); a structure that "
                               "was not built before the release is being asked for")
        return self.edge_index

    def _edge_subsets(self, group):
        """This rank's group's EdgeSubsets built from the ranks' slices, or None (option off, one rank, an emulated run)."""
        if not self.plan_from_slices or self.comm.world == 1 or self.comm.backend == "emulated":
            return None
        if group not in self._subsets:
            self._subsets[group] = subsets_from_slices(self._edges(), self.N_global, self.loops_mode, self.comm, group)
        return self._subsets[group]

    def _partition_plan(self, kind):
        sub = self._edge_subsets(1)
        if sub is not None:
            return PartitionPlan.from_subsets(sub, self.N_global, self.comm.world, self.comm.rank, kind)
        return PartitionPlan(self._edges(), self.N_global, self.comm.world, self.comm.rank, self.loops_mode, kind)

    def _grid_plan(self, kind, C, pieces):
        sub = self._edge_subsets(C)
        if sub is not None:
            return GridPlan.from_subsets(sub, self.N_global, self.comm.world, self.comm.rank, kind, C, pieces)
        return GridPlan(self._edges(), self.N_global, self.comm.world, self.comm.rank, self.loops_mode, kind, C, pieces)

    def release_edges(self):
        """Drop the global edge list: from here on only the structures already built (this rank's CSRs, send /
        receive lists, resident boundary rows) stay in HBM."""
        self.edge_index = None

    # ---- halo scheme ---------------------------------------------------------------------------
    def _get(self, kind):
        st = self._kinds.get(kind)
        if st is None:
            st = {"plan": self._partition_plan(kind)}
            self._kinds[kind] = st
        return st

    def _halo_csrs(self, kind, direction):
        st = self._get(kind)
        if direction not in st:
            half = getattr(st["plan"], direction)
            st[direction] = {
                "half": half,
                "loc": self.backend.prepare(half.loc_agg, half.loc_gather, half.n_local, half.loc_w),
                "rem": self.backend.prepare(half.rem_agg, half.rem_gather, half.n_local, half.rem_w)
                if half.n_halo else None,
            }
        return st[direction]

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
                recv, work = self.comm.all_to_all_rows(send, half.send_counts, half.recv_counts, tag="resident")
                work.wait()
                if half.n_halo:
                    ext = torch.cat([x, recv], dim=0)
            self._resident["ext"][key] = ext
        return ext

    def _ext_csr(self, kind):
        """CSR of this rank's targets over [local; halo] sources (+ per-edge weights in its slot order)."""
        st = self._get(kind)
        half = st["plan"].fwd
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
        return ops._PropagateLinear.apply(x_ext, _ExtGraph(csr, ws), "gcn", weight, bias, need_z, root_weight, x, False)

    def _run_halo(self, kind, direction, x):
        d = self._halo_csrs(kind, direction)
        half = d["half"]
        work = recv = None
        if self.comm.world > 1:
            send = self.backend.gather(x, half.send_idx) if half.n_send else x.new_empty((0, x.size(1)))
            recv, work = self.comm.all_to_all_rows(send, half.send_counts, half.recv_counts, tag="halo")
        out = self.backend.run(d["loc"], x, kind=f"dist_{direction}_local")  # overlaps the exchange
        if work is not None:
            work.wait()
            if d["rem"] is not None:
                out = self.backend.run(d["rem"], recv, y=out, kind=f"dist_{direction}_remote")
        return out

    # ---- grid schemes (reshard = 1 x P) ----------------------------------------------------------
    def _get_grid(self, kind, C, pieces):
        key = (kind, C, pieces)
        st = self._grid.get(key)
        if st is None:
            st = {"plan": self._grid_plan(kind, C, pieces)}
            self._grid[key] = st
        return st

    def _grid_half(self, kind, C, pieces, direction, src_pieces=1):
        """(GridHalf, CSR handle) of one direction — or, with `src_pieces` > 1, (GridHalf, [handle per source piece]):
        the edges are split by which row piece of its owner the SOURCE lies in (piece k of rank q = its rows
        [n_q k / n, n_q (k + 1) / n), the cut the inbound exchange of the fused schedule uses), so that the aggregation
        over the sources of piece k can run while piece k + 1 is still on the links."""
        if src_pieces <= 1:
            st = self._get_grid(kind, C, pieces)
            if direction not in st:
                half = getattr(st["plan"], direction)
                st[direction] = (half, self.backend.prepare(half.agg, half.gather, half.n_group, half.w))
                half.agg = half.gather = half.w = None  # the CSR holds them now
            return st[direction]
        key = (kind, C, pieces, "split", src_pieces)
        st = self._grid.get(key)
        if st is None:
            st = {"plan": self._grid_plan(kind, C, pieces)}
            self._grid[key] = st
        if direction not in st:
            half = getattr(st["plan"], direction)
            b = torch.tensor(self.bounds, dtype=half.gather.dtype, device=half.gather.device)
            owner = torch.bucketize(half.gather, b[1:], right=True)
            r = half.gather - b[owner]
            n_q = (b[1:] - b[:-1])[owner].clamp(min=1)
            piece = ((r + 1) * src_pieces - 1) // n_q
            handles = []
            for k in range(src_pieces):
                m = piece == k
                handles.append(self.backend.prepare(half.agg[m], half.gather[m], half.n_group,
                                                    None if half.w is None else half.w[m]))
            st[direction] = (half, handles)
            half.agg = half.gather = half.w = None
        return st[direction]

    def grid_plan(self, kind, d):
        """(GridPlan, R, C) the propagate of width d runs under, or None under the halo scheme."""
        shape = self.shape(d)
        if shape is None:
            return None
        return self._get_grid(kind, shape[1], self.pieces_for(d))["plan"], shape[0], shape[1]

    def _to_column_slice(self, x, R, C):
        """[n_local, d] row shard -> [N, d/C]: column slice c of EVERY node's row, c = this rank's slice. Every
        peer (r', c') is sent slice c' of my rows, so a slice travels to the R ranks that share it."""
        P, n_loc, dc = self.comm.world, self.n_local, x.size(1) // C
        send = x.view(n_loc, 1, C, dc).expand(n_loc, R, C, dc).permute(1, 2, 0, 3).reshape(P * n_loc, dc)
        cols, work = self.comm.all_to_all_rows(send, [n_loc] * P, self.row_counts, tag="in")
        work.wait()
        return cols

    def _aggregate_and_return(self, half, handle, cols, tag):
        """The row group's aggregate at this rank's column slice, returned to the owners of the rows, overlapped:
        piece k of every member's block is ONE contiguous row range of the piece-major CSR, aggregated by one launch
        straight into a send buffer and handed to an asynchronous all-to-all, which runs on RCCL's stream while piece
        k + 1 is aggregated. Only the last piece's exchange is exposed."""
        P, C, dc = self.comm.world, half.C, cols.size(1)
        run_rows = getattr(self.backend, "run_rows", None)
        out = cols.new_empty((half.n_local, C * dc))
        full = None
        pending = []
        for k in range(half.pieces):
            lo, hi = half.piece_ptr[k], half.piece_ptr[k + 1]
            send = None
            if full is None and run_rows is not None:
                send = cols.new_empty((hi - lo, dc))
                if run_rows(handle, cols, lo, hi, send, kind=tag) is False:
                    if pending:
                        raise RuntimeError("grid exchange: backend refused a row range after accepting one")
                    send = None
            if send is None:  # hub-row plan (row ids in it are absolute) or a backend without row ranges
                if full is None:
                    full = self.backend.run(handle, cols, kind=tag)
                send = full[lo:hi]
            a, b = half.my_piece[k]
            recv_counts = [b - a if q in half.members else 0 for q in range(P)]
            recv, work = self.comm.all_to_all_rows(send, half.piece_counts[k], recv_counts,
                                                   tag=f"out {k + 1}/{half.pieces}")
            pending.append((recv, work, a, b - a, send))  # `send` stays referenced until its exchange was waited on
        for recv, work, a, m, _send in pending:
            work.wait()
            if m:
                out[a:a + m].view(m, C, dc).copy_(recv.view(C, m, dc).permute(1, 0, 2))
        return out

    def _run_grid(self, kind, direction, x, shape):
        R, C = shape
        half, handle = self._grid_half(kind, C, self.pieces_for(x.size(1)), direction)
        cols = self._to_column_slice(x, R, C)
        return self._aggregate_and_return(half, handle, cols, f"dist_{direction}_colshard")

    def _appnp_columns(self, direction, h, K, alpha):
        """All K steps on the whole graph at width d/P between ONE pair of transposes (R = 1, one piece: the rows
        stay in node order, which the recurrence needs)."""
        P = self.comm.world
        half, handle = self._grid_half("gcn", P, 1, direction)
        cols = self._to_column_slice(h, 1, P)
        m = self.pieces_for(h.size(1)) if (K >= 1 and self.appnp_return_in_pieces) else 1
        if m > 1:
            # The LAST step's aggregation runs over the piece-major CSR and returns piece by piece, each piece's exchange
            # in flight while the next is aggregated (_aggregate_and_return, as every grid propagate): (m - 1) / m of the
            # outbound transpose leaves the critical path. The teleport term of that step is the owner's business:
            # z_K[own rows] = (1 - alpha) (A_hat z_{K-1})[own rows] + alpha h[own rows], on rows it holds anyway.
            self._appnp_return_pieces = m
            z = self.backend.appnp(handle, cols, K - 1, alpha, kind=f"dist_{direction}_appnp_colshard") if K > 1 else cols
            half_m, handle_m = self._grid_half("gcn", P, m, direction)
            ret = self._aggregate_and_return(half_m, handle_m, z, f"dist_{direction}_colshard")
            return torch.add(h * alpha, ret, alpha=1.0 - alpha)
        out = self.backend.appnp(handle, cols, K, alpha, kind=f"dist_{direction}_appnp_colshard")
        n_loc, dc = self.n_local, out.size(1)
        back, work = self.comm.all_to_all_rows(out, self.row_counts, [n_loc] * P, tag="out 1/1")
        work.wait()
        return back.view(P, n_loc, dc).permute(1, 0, 2).reshape(n_loc, P * dc)

    def appnp(self, h, K, alpha):
        """K-step APPNP on the partitioned graph; `None` when the per-iteration path should be used. The
        recurrence acts on every column independently, so under any column scheme the plain transpose (R = 1) with
        all K steps between its two exchanges moves the fewest bytes: 2 all-to-alls instead of 2K."""
        P = self.comm.world
        if P > 1 and self.exchange != "halo" and h.size(1) % P == 0 and self.shape(h.size(1)) is not None:
            return _DistAPPNPColumns.apply(h, self, K, alpha)
        return None

    # ---- choice --------------------------------------------------------------------------------
    # Cost model per propagate of width d (seconds; all ranks evaluate it on all-reduced inputs):
    #   gather rate  : the aggregation kernels sustain ~6.5 TB/s of 128-byte lines on this chip at every width
    #                  (a row of <= 32 floats costs one line), so
    #                  t_spmm = edges * ceil(4 * width / 128) * 128 B / 6.5e12
    #   link rate    : GB/s per direction per xGMI link, all P - 1 links concurrently: RGBX_LINK_GBS, else the rate
    #                  DistRunner measured at start-up with a 16 MB-per-peer all-to-all, else 60
    #   halo         : max(boundary bytes per link / rate, local-edge SpMM) + remote-edge SpMM
    #   grid R x C   : n_local * d/C * 4 B inbound per link, SpMM of E'/R edges at width d/C,
    #                  1/pieces of the outbound n_local * d/C * 4 B exposed
    GATHER_BPS = 6.5e12

    def _halo_stats(self):
        """(boundary rows this rank receives per forward propagate, its local-source edges, its remote-source
        edges) under the halo scheme."""
        if "halo_stats" not in self._choice:
            lo, hi = self.bounds[self.comm.rank], self.bounds[self.comm.rank + 1]
            sub = self._edge_subsets(1)
            if sub is not None:  # the edges into this rank's nodes are already here
                src = sub.by_dst[0]
                remote = (src < lo) | (src >= hi)
                n_mine = int(src.numel())
            else:
                src, dst = rewrite_global(self._edges(), self.N_global, self.loops_mode)
                mine = (dst >= lo) & (dst < hi)
                remote = mine & ((src < lo) | (src >= hi))
                n_mine = int(mine.sum())
            n_rem = int(remote.sum())
            self._choice["halo_stats"] = (int(torch.unique(src[remote]).numel()), n_mine - n_rem, n_rem)
        return self._choice["halo_stats"]

    def halo_rows(self):
        return self._halo_stats()[0]

    def costs(self, d):
        """{scheme name: modelled seconds per propagate of width d}, identical on every rank."""
        key = ("costs", d)
        if key in self._choice:
            return self._choice[key]
        import math
        import os
        P, N = self.comm.world, self.N_global
        # link rate: RGBX_LINK_GBS if set, else what Comm.measure_link_gbs() measured on this fabric, else 60 GB/s
        link = float(os.environ.get("RGBX_LINK_GBS") or getattr(self.comm, "link_gbs", None) or 60.0) * 1e9
        # what one small all-to-all costs end to end (Comm.measure_link_gbs; RGBX_LINK_LATENCY_US overrides, 0 when
        # nothing was measured): every exchange pays it once, so it decides how many pieces the outbound exchange is
        # worth cutting into
        lat = float(os.environ.get("RGBX_LINK_LATENCY_US") or getattr(self.comm, "link_latency_us", None) or 0.0) * 1e-6
        stats = torch.tensor([float(v) for v in self._halo_stats()], dtype=torch.float64,
                             device=self._edges().device)
        halo_rows, e_loc, e_rem = (self.comm.all_reduce_sum_(stats) / P).tolist()  # means over the ranks
        line_s = lambda edges, width: edges * math.ceil(4 * width / 128) * 128 / self.GATHER_BPS
        out = {}
        if self.exchange == "auto":
            out["halo"] = max(halo_rows * d * 4 / max(P - 1, 1) / link + lat, line_s(e_loc, d)) + line_s(e_rem, d)
        nnz = (e_loc + e_rem) * P
        self._choice["nnz"] = nnz
        from .plan import grid_shapes
        for R, C in grid_shapes(P):
            if C == 1 or d % C:
                continue  # C = 1 is the halo scheme with every remote row shipped
            name = "reshard" if R == 1 else f"grid{R}x{C}"
            per_link = (N / P) * (d / C) * 4 / link
            # p pieces: 1/p of the outbound bytes exposed, one more all-to-all latency and one more SpMM launch +
            # unpack copy (~20 us of rank compute) per piece
            options = [self.pieces] if self._pieces_fixed else [1, 2, 4, 8]
            tail = {p: per_link / p + (1 + p) * lat + 20e-6 * p for p in options}
            best_p = min(tail, key=lambda p: (tail[p], p))
            self._choice[("pieces", d, name)] = best_p
            out[name] = per_link + line_s(nnz / R, d // C) + tail[best_p]
        self._choice[key] = out
        return out

    def replicate_costs(self, d_in, d_h):
        """Modelled seconds per EPOCH (training forward + backward, two eval forwards) of a model's first two conv
        layers (input width d_in, hidden width d_h), identical on every rank: {"exchange": first layer on the resident
        features at 1/P + the second layer's four exchanged propagates under the best scheme, "replicate": first layer
        on all N rows by every rank + the second layer's rectangular aggregation without any exchange (ReplicaGraph)}.
        The dense work on the hidden matrix (BatchNorm, weight gradients: ~10 passes per epoch at ~5 TB/s) is
        counted at N rows vs N/P."""
        import math
        P, N = self.comm.world, self.N_global
        best = min(self.costs(d_h).values())
        nnz = self._choice["nnz"]
        line_s = lambda edges, width: edges * math.ceil(4 * width / 128) * 128 / self.GATHER_BPS
        dense = N * d_h * 4 * 10 / 5e12
        return {"exchange": 3 * line_s(nnz / P, d_in) + 4 * best + dense / P,
                "replicate": 3 * line_s(nnz, d_in) + 4 * line_s(nnz / P, d_h) + dense}

    def pieces_for(self, d):
        """Pieces the outbound exchange of a width-d propagate is cut into: the caller's, else the cost model's pick
        for the scheme in use (identical on every rank)."""
        if self._pieces_fixed or self.comm.world == 1 or self.exchange == "halo":
            return self.pieces
        self.costs(d)
        n = self._choice.get(("pieces", d, self.scheme(d)), self.pieces)
        # The cost model prices the last outbound piece as exposed. A schedule that interleaves other work with the
        # exchanges (dist/stack.py: the eval pair, the step computed ahead) hides it anyway and only pays for the extra
        # launches and exchanges: `auto_pieces_cap` (set by DistRunner for that schedule; measured: DESIGN.md 4.4)
        cap = getattr(self, "auto_pieces_cap", None)
        return n if cap is None else min(n, cap)

    def shape(self, d):
        """(R, C) of the grid scheme a propagate of width d uses, or None for the halo scheme."""
        P = self.comm.world
        if P == 1 or self.exchange == "halo":
            return None
        if self.exchange == "reshard":
            return (1, P) if d % P == 0 else None
        if "x" in self.exchange:  # explicit "RxC"
            R, C = (int(v) for v in self.exchange.split("x"))
            if R * C != P:
                raise ValueError(f"exchange={self.exchange!r} does not factor the world size {P}")
            return (R, C) if C > 1 and d % C == 0 else None
        costs = self.costs(d)
        best = min(costs, key=lambda k: (costs[k], k))
        if best == "halo":
            return None
        if best == "reshard":
            return (1, P)
        R, C = (int(v) for v in best[4:].split("x"))
        return (R, C)

    def scheme(self, d):
        """'halo', 'reshard' (1 x P) or 'gridRxC' for feature width d (all ranks reach the same answer)."""
        shape = self.shape(d)
        if shape is None:
            return "halo"
        return "reshard" if shape[0] == 1 else f"grid{shape[0]}x{shape[1]}"

    def _run(self, kind, direction, x):
        if direction == "fwd" and self.is_resident(x):
            return self._run_resident(kind, x)
        shape = self.shape(x.size(1)) if self.comm.world > 1 else None
        if shape is not None:
            return self._run_grid(kind, direction, x.contiguous(), shape)
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
            plan = self._partition_plan("sum")
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


class _TailGraph:
    """What ops._PropagateLinear / spmm_raw read of the rectangular graph [this rank's targets x all N sources]."""
    event_prefix = "tail_"

    def __init__(self, fwd, bwd):
        (self.fwd, self.w), (self.bwd, self.w_t) = fwd, bwd
        self.inv_deg = None


class _FullGraph:
    event_prefix = "replica_"

    def __init__(self, fwd):
        self.fwd, self.w = fwd
        self.inv_deg = None


class _ReplicaPropagate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rgraph, kind, full):
        ctx.rgraph, ctx.kind, ctx.full = rgraph, kind, full
        st = rgraph._build(kind)
        if full:
            return rgraph.dg.backend.run(st["full"], x, kind="replica_fwd")
        return rgraph.dg.backend.run(st["tail_fwd"], x.contiguous(), kind="tail_fwd")

    @staticmethod
    def backward(ctx, gy):
        if ctx.full:
            raise RuntimeError("ReplicaGraph: the replicated first layer aggregates the static input features, which "
                               "take no gradient")
        st = ctx.rgraph._build(ctx.kind)
        return ctx.rgraph.dg.backend.run(st["tail_bwd"], gy.contiguous(), kind="tail_bwd"), None, None, None


class ReplicaGraph:
    """Exchange-free scheme for the first two conv layers ("replicate"): every rank holds the whole static feature
    matrix and computes the FIRST conv layer (and the BatchNorm behind it) for ALL N nodes — the single-GPU work,
    redundantly — and the SECOND conv layer only for its own targets, over a rectangular CSR [n_local x N] whose
    sources are those replicated rows. No activation row ever crosses a link; the only collectives left are the
    small all-reduces (loss, parameter gradients). Backward: the rectangular transposed SpMM gives every rank the
    PARTIAL gradient of the replicated hidden rows that stems from its own targets; BatchNorm's and the first
    layer's backward are linear in that gradient, so the partial parameter gradients add up to the true ones in the
    gradient all-reduce that runs anyway (DistBatchNorm1d._reducer).

    Costs (1 - 1/P) of the first layer's full-graph aggregation extra per forward and saves every exchange of the
    second layer: the scheme of choice where the exchange is dearer than that — 2 GPUs (one xGMI link per pair), or
    any world size when the measured link rate is low (DistGraph.replicate_pays). Registered for the conv layers
    under (token, N_global rows); which of the two stages a call is follows from its input: the pinned feature
    matrix itself -> first layer, any other N-row tensor -> second layer (so the first conv must aggregate the
    features BEFORE transforming them: in_channels <= out_channels, which DistRunner checks)."""

    is_distributed = True

    def __init__(self, dgraph):
        self.dg = dgraph
        self.comm, self.backend = dgraph.comm, dgraph.backend
        self._st = {}
        self._x = None
        self.lo = dgraph.bounds[dgraph.comm.rank]
        self.hi = self.lo + dgraph.n_local

    def pin_resident(self, x_full):
        self._x = (x_full.data_ptr(), tuple(x_full.shape), x_full._version)

    def is_resident(self, x):
        return (self._x is not None and not x.requires_grad
                and (x.data_ptr(), tuple(x.shape), x._version) == self._x)

    def _build(self, kind):
        st = self._st.get(kind)
        if st is None:
            dg = self.dg
            src, dst = rewrite_global(dg._edges(), dg.N_global, dg.loops_mode)
            from .plan import edge_weights
            w = edge_weights(src, dst, dg.N_global, kind)
            mine = (dst >= self.lo) & (dst < self.hi)
            wm = None if w is None else w[mine]
            be = dg.backend
            st = {"full": be.prepare(dst, src, dg.N_global, w),
                  "tail_fwd": be.prepare(dst[mine] - self.lo, src[mine], dg.n_local, wm),
                  "tail_bwd": be.prepare(src[mine], dst[mine] - self.lo, dg.N_global, wm),
                  "nnz_total": int(src.numel()), "nnz_tail": int(mine.sum())}
            self._st[kind] = st
        return st

    def release_edges(self):
        pass  # the edge list lives in the DistGraph this object was made from

    def target_rows(self, x):
        """Rows of a conv input that are targets of its aggregation: all N for the first stage, mine for the second."""
        return x if self.is_resident(x) or x.size(0) != self.dg.N_global else x[self.lo:self.hi]

    def propagate(self, x, kind):
        return _ReplicaPropagate.apply(x, self, kind, self.is_resident(x))

    def fused_resident_ok(self, x, in_channels, out_channels, root):
        from .. import _lib
        if not (x.is_cuda and isinstance(self.backend, HipAggregator)) or x.size(0) != self.dg.N_global:
            return False
        # the tail adds its root term outside the kernel (the targets' own rows are a row range of the sources)
        return bool(_lib.load().rgbx_spmm_linear_supported(in_channels, out_channels,
                                                           int(root and self.is_resident(x))))

    def propagate_linear(self, x, kind, weight, bias, root_weight):
        from .. import ops
        st = self._build(kind)
        need_z = torch.is_grad_enabled() and weight.requires_grad
        if self.is_resident(x):  # first layer, all N rows: the single-GPU launch
            return ops._PropagateLinear.apply(x, _FullGraph(st["full"]), "gcn", weight, bias, need_z, root_weight,
                                              None, False)
        out = ops._PropagateLinear.apply(x, _TailGraph(st["tail_fwd"], st["tail_bwd"]), "gcn", weight, bias, need_z,
                                         None, None, False)
        if root_weight is not None:
            out = out + ops.linear(x[self.lo:self.hi], root_weight)
        return out

    def appnp(self, h, K, alpha):
        raise RuntimeError("ReplicaGraph serves the conv-stack models only")


def install(token_edge_index, n_local, edge_index, num_nodes, comm=None, backend=None, exchange="auto", pieces=None):
    """Register DistGraphs so that conv layers called with (x_local, token_edge_index) aggregate over
    the partitioned global graph. Returns {loops_mode: DistGraph}."""
    graphs = {}
    for mode in (_graph.LOOPS_KEEP, _graph.LOOPS_ADD_REMAINING, _graph.LOOPS_REMOVE_ADD):
        g = DistGraph(edge_index, num_nodes, mode, comm, backend, "auto" if exchange == "replicate" else exchange,
                      pieces)
        _graph.register_graph(token_edge_index, n_local, mode, g)
        graphs[mode] = g
    return graphs


def install_replicas(token_edge_index, graphs, num_nodes):
    """Register a ReplicaGraph per rewrite mode for inputs of N_global rows (the replicate scheme). Returns
    {loops_mode: ReplicaGraph}."""
    out = {}
    for mode, g in graphs.items():
        r = ReplicaGraph(g)
        _graph.register_graph(token_edge_index, num_nodes, mode, r)
        out[mode] = r
    return out
