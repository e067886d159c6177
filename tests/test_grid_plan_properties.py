"""Property tests (hypothesis) of the row-group x column-slice plan on random graphs, worlds and piece counts: CPU only.
The plan is pure index arithmetic; what must hold is that an emulation of the whole exchange with plain tensors —
column slices in, per-group aggregation in piece-major row order, pieces back to the row owners — reproduces the
single-process propagate for every rank."""
import os

import torch
from hypothesis import given, settings, strategies as st

from oracle import ref_cpu as O
from rgb_experiment_amd.dist.plan import GridPlan, partition_bounds


@st.composite
def cases(draw):
    world_c = draw(st.sampled_from([(2, 2), (4, 2), (4, 4), (6, 3), (6, 2), (3, 3), (8, 4)]))
    n = draw(st.integers(min_value=world_c[0], max_value=60))
    e = draw(st.integers(min_value=0, max_value=300))
    pieces = draw(st.integers(min_value=1, max_value=5))
    seed = draw(st.integers(min_value=0, max_value=10_000))
    mode = draw(st.sampled_from([0, 1, 2]))
    return world_c, n, e, pieces, seed, mode


@settings(max_examples=40, deadline=None)
@given(cases())
def test_grid_exchange_emulated_with_tensors_equals_the_propagate(case):
    (world, C), n, e, pieces, seed, mode = case
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, n, (2, e), generator=g)
    d = C * 3
    x = torch.randn(n, d, generator=g)
    kind = "gcn" if mode == 1 else "mean"
    rei, _ = O.rewrite_edges(ei, n, mode)
    if kind == "gcn":
        _, w = O.gcn_norm(ei, None, n)
        want = O.propagate(rei, x, n, w, "add")
    else:
        want = O.propagate(rei, x, n, None, "mean")
    b = partition_bounds(n, world)
    dc = d // C
    plans = [GridPlan(ei, n, world, r, mode, kind, C, pieces) for r in range(world)]
    out = torch.zeros(n, d)
    for p, plan in enumerate(plans):
        h = plan.fwd
        r, c = p // C, p % C
        cols = x[:, c * dc:(c + 1) * dc]                      # what the inbound all-to-all assembles on rank p
        ei_g = torch.stack([h.gather, h.agg])
        agg = O.propagate(ei_g, cols, h.n_group, h.w, "add")  # rows in piece-major order
        for k in range(pieces):                               # piece k back to the owners of its rows
            off = h.piece_ptr[k]
            for q in h.members:
                cnt = h.piece_counts[k][q]
                a = (b[q + 1] - b[q]) * k // pieces
                out[b[q] + a:b[q] + a + cnt, c * dc:(c + 1) * dc] = agg[off:off + cnt]
                off += cnt
            assert off == h.piece_ptr[k + 1]
    assert torch.allclose(out, want, atol=1e-5)


def test_replicate_cost_model_follows_the_link_rate(monkeypatch):
    """DistGraph.replicate_costs (what DistRunner's "auto" compares): with links that deliver nothing the replicated
    first layer must win, with free links the exchange schemes must; both figures are positive and identical on every
    rank by construction (inputs are all-reduced)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _dist_worker import OracleAggregator
    from rgb_experiment_amd.dist.comm import EmulatedComm
    from rgb_experiment_amd.dist.graph import DistGraph
    g = torch.Generator().manual_seed(7)
    n = 4000
    ei = torch.randint(0, n, (2, 60000), generator=g)
    picks = {}
    for gbs in ("0.000001", "1000000"):
        monkeypatch.setenv("RGBX_LINK_GBS", gbs)
        for world in (2, 8):
            dg = DistGraph(ei, n, 1, EmulatedComm(world, 0), OracleAggregator(), "auto")
            c = dg.replicate_costs(128, 128)
            assert set(c) == {"exchange", "replicate"} and min(c.values()) > 0
            picks[(gbs, world)] = "replicate" if c["replicate"] < c["exchange"] else "exchange"
    assert picks[("0.000001", 2)] == picks[("0.000001", 8)] == "replicate"
    assert picks[("1000000", 2)] == picks[("1000000", 8)] == "exchange"


def test_powerlaw_endpoints_are_skewed_and_in_range():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    ids = bench.powerlaw_endpoints(5000, 200000, 3)
    assert ids.dtype == torch.int64 and int(ids.min()) >= 0 and int(ids.max()) < 5000
    deg = torch.bincount(ids, minlength=5000)
    assert int(deg.max()) > 20 * int(deg.float().median())  # a hub: 200000 / sqrt(5000) ~ 2800 against a median of ~20
    assert torch.equal(ids, bench.powerlaw_endpoints(5000, 200000, 3))  # seeded


def test_piece_count_follows_latency_and_link_rate(monkeypatch):
    """Outbound pieces when the caller fixes none: slow links with free exchanges -> many pieces (less of the transfer
    exposed), costly exchanges on fast links -> one; a caller's count is kept."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _dist_worker import OracleAggregator
    from rgb_experiment_amd.dist.comm import EmulatedComm
    from rgb_experiment_amd.dist.graph import DistGraph
    g = torch.Generator().manual_seed(11)
    n = 4000
    ei = torch.randint(0, n, (2, 60000), generator=g)
    mk = lambda **kw: DistGraph(ei, n, 1, EmulatedComm(8, 0), OracleAggregator(), "2x4", **kw)
    monkeypatch.setenv("RGBX_LINK_GBS", "0.0001")
    monkeypatch.setenv("RGBX_LINK_LATENCY_US", "0")
    assert mk().pieces_for(128) == 8
    monkeypatch.setenv("RGBX_LINK_GBS", "100000")
    monkeypatch.setenv("RGBX_LINK_LATENCY_US", "500")
    assert mk().pieces_for(128) == 1
    assert mk(pieces=3).pieces_for(128) == 3
