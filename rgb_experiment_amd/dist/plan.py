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


class HalfPlan:
    """One direction (forward: agg = target, gather = source; backward: roles swapped) for one rank."""

    def __init__(self, agg, gather, weight, num_nodes, world, rank):
        bounds = partition_bounds(num_nodes, world)
        lo, hi = bounds[rank], bounds[rank + 1]
        dev = agg.device
        self.world, self.rank, self.lo, self.hi = world, rank, lo, hi
        self.n_local = hi - lo
        b = torch.tensor(bounds[1:], dtype=agg.dtype, device=dev)

        mine = (agg >= lo) & (agg < hi)
        a = agg[mine] - lo
        g = gather[mine]
        w = None if weight is None else weight[mine]
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
        theirs = (gather >= lo) & (gather < hi) & ~mine
        q_of_edge = torch.bucketize(agg[theirs], b, right=True)
        key = torch.unique(q_of_edge * (hi - lo if hi > lo else 1) + (gather[theirs] - lo), sorted=True)
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

    def __init__(self, agg, gather, weight, num_nodes, world, rank, C, pieces):
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
