"""1-D contiguous node partition of a graph for full-graph training on P GPUs (new capability: the
reference is single-device, itexperiments.py:246).

Rank p owns nodes [bounds[p], bounds[p+1]): their feature rows, outputs, labels, and every edge that
AGGREGATES INTO one of them. Edges are split by where they gather from:
  * local edges  — source owned by p: run on the local feature rows while the exchange is in flight;
  * remote edges — source owned by a peer q: read from a compact halo buffer that holds, per peer in
    rank order, the sorted unique rows p needs from q.
The same plan built with the roles of source and target swapped serves backward (dX = A^T dY): rank p
then owns the edges LEAVING its nodes and the halo carries dY rows.

Everything here is index arithmetic in plain torch ops (device-agnostic, so it is unit-tested on CPU);
the heavy grouping (CSR build) and all arithmetic on features happen in the HIP library.
"""
import torch


def partition_bounds(num_nodes, world):
    """Contiguous, near-equal node ranges: bounds[p] = p * N // P."""
    return [p * num_nodes // world for p in range(world + 1)]


def rewrite_global(edge_index, num_nodes, loops_mode):
    """The edge list a conv layer aggregates over (same rule as rgbx_csr_build / reference
    models/dagnn.py:20-24, models/graphsage.py:53-56): for loops_mode != 0 drop self-loops, keep the
    other edges in order, append one self-loop per node."""
    src, dst = edge_index[0], edge_index[1]
    if loops_mode == 0:
        return src, dst
    keep = src != dst
    loops = torch.arange(num_nodes, dtype=edge_index.dtype, device=edge_index.device)
    return torch.cat([src[keep], loops]), torch.cat([dst[keep], loops])


def edge_weights(src, dst, num_nodes, kind):
    """Per rewritten edge weight. 'gcn': dis[src]*dis[dst] with dis = in-degree^-1/2, inf -> 0
    (models/dagnn.py:26-31); 'mean': 1/max(in-degree,1) of the target (aggr='mean'); 'sum': None."""
    if kind == "sum":
        return None
    deg = torch.bincount(dst, minlength=num_nodes).to(torch.float32)
    if kind == "gcn":
        dis = deg.pow(-0.5)
        dis.masked_fill_(dis == float("inf"), 0)
        return dis[src] * dis[dst]
    if kind == "mean":
        return (1.0 / deg.clamp(min=1))[dst]
    raise ValueError(kind)


def subset_weights(src, dst, deg, kind):
    """edge_weights for a SUBSET of the rewritten edges, from the global in-degree vector `deg` (int64 [N]): the same
    float32 operations per edge, so the same bits."""
    if kind == "sum":
        return None
    deg = deg.to(torch.float32)
    if kind == "gcn":
        dis = deg.pow(-0.5)
        dis.masked_fill_(dis == float("inf"), 0)
        return dis[src] * dis[dst]
    if kind == "mean":
        return (1.0 / deg.clamp(min=1))[dst]
    raise ValueError(kind)


class EdgeSubsets:
    """What the ranks of one group of `group` consecutive ranks need of the rewritten edge list to build their plans: the
    edges whose TARGET lies in the group's node range (`by_dst`) and those whose SOURCE does (`by_src`), each as (src, dst)
    in the order of the global rewritten list, the global in-degree of the rewritten list and its length."""

    def __init__(self, by_dst, by_src, deg, nnz_total, lo, hi):
        self.by_dst, self.by_src, self.deg, self.nnz_total, self.lo, self.hi = by_dst, by_src, deg, nnz_total, lo, hi


def subsets_from_global(edge_index, num_nodes, loops_mode, lo, hi):
    """EdgeSubsets of the node range [lo, hi) from the WHOLE edge list (every rank scans all E edges)."""
    src, dst = rewrite_global(edge_index, num_nodes, loops_mode)
    t, s = (dst >= lo) & (dst < hi), (src >= lo) & (src < hi)
    return EdgeSubsets((src[t], dst[t]), (src[s], dst[s]), torch.bincount(dst, minlength=num_nodes), int(src.numel()), lo, hi)


def subsets_from_slices(edge_index, num_nodes, loops_mode, comm, group=1):
    """The same EdgeSubsets, every rank touching only ITS 1/P of the edge list (edges [rank E / P, (rank + 1) E / P)):
    drop the slice's self-loops (loops_mode != 0), bucket it by the group that owns the target resp. the source (stable: the
    slice's order survives inside a bucket), one all-to-all of (src, dst) records per direction — a rank receives its
    buckets in rank order, i.e. in the order of the global list, since the slices are consecutive ranges of it — and append
    the group's own self-loops, which the global rewrite puts at the end in node order. The in-degree is the all-reduced
    bincount of the slices' targets. Collective: every rank of `comm` calls it with the same arguments."""
    world, rank = comm.world, comm.rank
    if world % group:
        raise ValueError(f"group size {group} does not divide the world size {world}")
    E = edge_index.size(1)
    a, b = rank * E // world, (rank + 1) * E // world
    src, dst = edge_index[0, a:b], edge_index[1, a:b]
    if loops_mode != 0:
        keep = src != dst
        src, dst = src[keep], dst[keep]
    dev = edge_index.device
    bounds = partition_bounds(num_nodes, world)
    n_groups = world // group
    upper = torch.tensor(bounds[group::group], dtype=edge_index.dtype, device=dev)  # upper bound of every group's range
    deg = torch.bincount(dst, minlength=num_nodes)
    total = torch.tensor([src.numel()], dtype=torch.int64, device=dev)
    comm.all_reduce_sum_(deg)
    comm.all_reduce_sum_(total)
    nnz_total = int(total.item())
    my_group = rank // group
    lo, hi = bounds[my_group * group], bounds[(my_group + 1) * group]
    if loops_mode != 0:
        deg = deg + 1
        nnz_total += num_nodes
    loops = torch.arange(lo, hi, dtype=edge_index.dtype, device=dev)

    def route(key):
        owner = torch.bucketize(key, upper, right=True)
        order = torch.argsort(owner, stable=True)
        per_group = torch.bincount(owner, minlength=n_groups)
        rec = torch.stack([src[order], dst[order]], dim=1)  # [n, 2] records, grouped by owning group
        send_counts = [int(c) for c in per_group.tolist() for _ in range(group)]  # every member of a group gets its bucket
        if group > 1:
            ptr = [0]
            for c in per_group.tolist():
                ptr.append(ptr[-1] + int(c))
            rec = torch.cat([rec[ptr[q // group]:ptr[q // group + 1]] for q in range(world)], dim=0)
        counts = torch.zeros((world, world), dtype=torch.int64, device=dev)
        counts[rank] = torch.tensor(send_counts, dtype=torch.int64, device=dev)
        comm.all_reduce_sum_(counts)
        recv_counts = counts[:, rank].tolist()
        got, work = comm.all_to_all_rows(rec, send_counts, recv_counts, tag="edge records")
        work.wait()
        if world == 1:
            got = rec
        s, d = got[:, 0], got[:, 1]
        if loops_mode != 0:
            s, d = torch.cat([s, loops]), torch.cat([d, loops])
        return s.contiguous(), d.contiguous()

    return EdgeSubsets(route(dst), route(src), deg, nnz_total, lo, hi)


class HalfPlan:
    """One direction (forward: agg = target, gather = source; backward: roles swapped) for one rank."""

    def __init__(self, agg, gather, weight, num_nodes, world, rank):
        """From the GLOBAL rewritten list (agg, gather, weight: one entry per edge)."""
        bounds = partition_bounds(num_nodes, world)
        lo, hi = bounds[rank], bounds[rank + 1]
        mine = (agg >= lo) & (agg < hi)
        theirs = (gather >= lo) & (gather < hi) & ~mine
        self._build(agg[mine], gather[mine], None if weight is None else weight[mine], agg[theirs], gather[theirs],
                    num_nodes, world, rank)

    @classmethod
    def from_subsets(cls, agg_mine, gather_mine, w_mine, agg_gathered_here, gather_here, num_nodes, world, rank):
        """From the two subsets a rank holds (EdgeSubsets): the edges that aggregate into its nodes, and the edges that gather
        from its nodes (of which those aggregating elsewhere decide what it sends), each in global order."""
        self = cls.__new__(cls)
        bounds = partition_bounds(num_nodes, world)
        lo, hi = bounds[rank], bounds[rank + 1]
        elsewhere = (agg_gathered_here < lo) | (agg_gathered_here >= hi)
        self._build(agg_mine, gather_mine, w_mine, agg_gathered_here[elsewhere], gather_here[elsewhere], num_nodes, world,
                    rank)
        return self

    def _build(self, agg_mine, g, w, agg_theirs, gather_theirs, num_nodes, world, rank):
        bounds = partition_bounds(num_nodes, world)
        lo, hi = bounds[rank], bounds[rank + 1]
        dev = agg_mine.device
        self.world, self.rank, self.lo, self.hi = world, rank, lo, hi
        self.n_local = hi - lo
        b = torch.tensor(bounds[1:], dtype=agg_mine.dtype, device=dev)
        a = agg_mine - lo
        is_local = (g >= lo) & (g < hi)

        self.loc_agg, self.loc_gather = a[is_local], g[is_local] - lo
        self.loc_w = None if w is None else w[is_local]

        g_rem = g[~is_local]
        halo_ids = torch.unique(g_rem, sorted=True)  # contiguous ranges => already grouped by owner
        self.halo_ids = halo_ids
        self.n_halo = int(halo_ids.numel())
        self.rem_agg = a[~is_local]
        self.rem_gather = torch.searchsorted(halo_ids, g_rem)
        self.rem_w = None if w is None else w[~is_local]
        owner = torch.bucketize(halo_ids, b, right=True)
        self.recv_counts = torch.bincount(owner, minlength=world).tolist()

        # rows of mine that each peer q needs = unique gather ids in my range among q's edges
        q_of_edge = torch.bucketize(agg_theirs, b, right=True)
        key = torch.unique(q_of_edge * (hi - lo if hi > lo else 1) + (gather_theirs - lo), sorted=True)
        span = max(hi - lo, 1)
        self.send_idx = (key % span).to(torch.int32)
        self.send_counts = torch.bincount(torch.div(key, span, rounding_mode="floor"), minlength=world).tolist()

    @property
    def n_send(self):
        return int(self.send_idx.numel())


class PartitionPlan:
    """Forward + backward halves for one (graph, rewrite mode, weighting kind, rank)."""

    def __init__(self, edge_index, num_nodes, world, rank, loops_mode, kind):
        src, dst = rewrite_global(edge_index, num_nodes, loops_mode)
        w = edge_weights(src, dst, num_nodes, kind)
        self.num_nodes, self.world, self.rank, self.kind = num_nodes, world, rank, kind
        self.nnz_total = int(src.numel())
        self.fwd = HalfPlan(dst, src, w, num_nodes, world, rank)
        self.bwd = HalfPlan(src, dst, w, num_nodes, world, rank)
        self.n_local = self.fwd.n_local
        self.nnz_local = int(self.fwd.loc_agg.numel() + self.fwd.rem_agg.numel())

    @classmethod
    def from_subsets(cls, sub, num_nodes, world, rank, kind):
        """The same plan from this rank's EdgeSubsets (group size 1): no pass over the whole edge list."""
        self = cls.__new__(cls)
        (ts, td), (ss, sd) = sub.by_dst, sub.by_src
        self.num_nodes, self.world, self.rank, self.kind = num_nodes, world, rank, kind
        self.nnz_total = sub.nnz_total
        self.fwd = HalfPlan.from_subsets(td, ts, subset_weights(ts, td, sub.deg, kind), sd, ss, num_nodes, world, rank)
        self.bwd = HalfPlan.from_subsets(ss, sd, subset_weights(ss, sd, sub.deg, kind), ts, td, num_nodes, world, rank)
        self.n_local = self.fwd.n_local
        self.nnz_local = int(self.fwd.loc_agg.numel() + self.fwd.rem_agg.numel())
        return self


def grid_shapes(world):
    """(R, C) factorisations of the world size: R row groups x C column slices."""
    return [(world // c, c) for c in range(1, world + 1) if world % c == 0]


class GridHalf:
    """One direction of the row-group x column-slice scheme for one rank.

    Rank p = r*C + c. Its ROW GROUP r is the union of the node ranges of ranks r*C .. r*C+C-1; its COLUMN SLICE c
    is columns [c*d/C, (c+1)*d/C) of whatever feature width d is propagated. Per propagate the rank
      1. receives column slice c of EVERY node's row (all-to-all, every peer sends its own rows),
      2. aggregates into the rows of its row group from all N sources at width d/C,
      3. returns to each rank of its row group that rank's rows (all-to-all inside the row group).
    R = 1 is the plain transpose ("reshard"); R > 1 keeps d/C at a full 128-byte line for narrow d/P at the price
    of a larger inbound exchange.

    The group's rows are stored PIECE-MAJOR: piece k of every group member's block first (members in rank order),
    then piece k+1, ... so that the rows one outbound all-to-all carries form ONE contiguous row range of the CSR,
    already in send-buffer order. `pieces` is fixed here.
    agg / gather: GLOBAL int64 index vectors of the rewritten edge list (aggregate-into, gather-from)."""

    def __init__(self, agg, gather, weight, num_nodes, world, rank, C, pieces, prefiltered=False):
        """`prefiltered`: agg / gather / weight hold only the edges that aggregate into this rank's row group (EdgeSubsets of
        group size C), in global order."""
        bounds = partition_bounds(num_nodes, world)
        R = world // C
        r, c = rank // C, rank % C
        self.R, self.C, self.r, self.c, self.pieces = R, C, r, c, pieces
        self.members = list(range(r * C, (r + 1) * C))  # ranks of my row group
        glo, ghi = bounds[r * C], bounds[(r + 1) * C]
        self.group_lo, self.n_group = glo, ghi - glo
        self.n_local = bounds[rank + 1] - bounds[rank]
        self.row_counts = [bounds[q + 1] - bounds[q] for q in range(world)]
        dev = agg.device

        cut = lambda n, k: (n * k) // pieces
        # position of every group row in piece-major order, and per piece the rows each member receives
        pos = torch.empty(self.n_group, dtype=torch.int64, device=dev)
        self.piece_ptr = [0]
        self.piece_counts = []  # [pieces][world] rows sent to every rank in piece k (0 outside the row group)
        off = 0
        for k in range(pieces):
            counts = [0] * world
            for q in self.members:
                nq = bounds[q + 1] - bounds[q]
                a, b = cut(nq, k), cut(nq, k + 1)
                lo = bounds[q] - glo
                pos[lo + a:lo + b] = torch.arange(off, off + (b - a), device=dev)
                off += b - a
                counts[q] = b - a
            self.piece_counts.append(counts)
            self.piece_ptr.append(off)
        # rows of mine that piece k returns: [cut(n_local, k), cut(n_local, k+1))
        self.my_piece = [(cut(self.n_local, k), cut(self.n_local, k + 1)) for k in range(pieces)]

        if prefiltered:
            self.agg, self.gather, self.w = pos[agg - glo], gather, weight
        else:
            mine = (agg >= glo) & (agg < ghi)
            self.agg = pos[agg[mine] - glo]
            self.gather = gather[mine]
            self.w = None if weight is None else weight[mine]
        self.nnz = int(self.agg.numel())


class GridPlan:
    """Forward + backward GridHalf for one (graph, rewrite mode, weighting kind, rank, C, pieces)."""

    def __init__(self, edge_index, num_nodes, world, rank, loops_mode, kind, C, pieces):
        src, dst = rewrite_global(edge_index, num_nodes, loops_mode)
        w = edge_weights(src, dst, num_nodes, kind)
        self.num_nodes, self.world, self.rank, self.kind = num_nodes, world, rank, kind
        self.nnz_total = int(src.numel())
        self.fwd = GridHalf(dst, src, w, num_nodes, world, rank, C, pieces)
        self.bwd = GridHalf(src, dst, w, num_nodes, world, rank, C, pieces)

    @classmethod
    def from_subsets(cls, sub, num_nodes, world, rank, kind, C, pieces):
        """The same plan from the EdgeSubsets of this rank's row group (group size C)."""
        self = cls.__new__(cls)
        (ts, td), (ss, sd) = sub.by_dst, sub.by_src
        self.num_nodes, self.world, self.rank, self.kind = num_nodes, world, rank, kind
        self.nnz_total = sub.nnz_total
        self.fwd = GridHalf(td, ts, subset_weights(ts, td, sub.deg, kind), num_nodes, world, rank, C, pieces, prefiltered=True)
        self.bwd = GridHalf(ss, sd, subset_weights(ss, sd, sub.deg, kind), num_nodes, world, rank, C, pieces, prefiltered=True)
        return self
