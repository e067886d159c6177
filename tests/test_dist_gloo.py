"""world_size 2 and 3 on CPU (gloo): the 1-D node partition, the all-to-all halo exchange, global
BatchNorm statistics and gradient all-reduce reproduce single-process results."""
import copy
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import _dist_worker as W
from oracle import ref_cpu as O


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_plan_is_consistent_and_covers_every_edge():
    from rgb_experiment_amd.dist import PartitionPlan, partition_bounds
    ei, x, _, _ = W.make_problem()
    n = x.size(0)
    for world in (1, 2, 3, 5):
        plans = [PartitionPlan(ei, n, world, r, 1, "gcn") for r in range(world)]
        assert sum(p.nnz_local for p in plans) == plans[0].nnz_total
        b = partition_bounds(n, world)
        assert b[0] == 0 and b[-1] == n
        for p in range(world):
            for half in ("fwd", "bwd"):
                hp = getattr(plans[p], half)
                assert sum(hp.recv_counts) == hp.n_halo and hp.recv_counts[p] == 0
                assert sum(hp.send_counts) == hp.n_send and hp.send_counts[p] == 0
                off = 0
                for q in range(world):
                    hq = getattr(plans[q], half)
                    assert hp.send_counts[q] == hq.recv_counts[p]
                    # the rows p sends to q are exactly the rows q expects from p, in the same order
                    q_off = sum(hq.recv_counts[:p])
                    want = hq.halo_ids[q_off:q_off + hq.recv_counts[p]]
                    got = hp.send_idx[off:off + hp.send_counts[q]].long() + hp.lo
                    assert torch.equal(got, want)
                    off += hp.send_counts[q]
                if hp.rem_gather.numel():
                    assert int(hp.rem_gather.max()) < hp.n_halo
                if hp.loc_gather.numel():
                    assert int(hp.loc_gather.max()) < hp.n_local


def test_grid_plan_covers_every_edge_once_and_orders_rows_piece_major():
    """plan.GridHalf: the row groups partition the edges; the piece-major order is a permutation of the group's
    rows in which piece k of every member's block is one contiguous range."""
    from rgb_experiment_amd.dist.plan import GridPlan, grid_shapes, partition_bounds
    ei, x, _, _ = W.make_problem()
    n = x.size(0)
    assert grid_shapes(8) == [(8, 1), (4, 2), (2, 4), (1, 8)]
    for world, C, pieces in ((4, 2, 3), (6, 3, 2), (6, 2, 4), (4, 4, 1)):
        b = partition_bounds(n, world)
        plans = [GridPlan(ei, n, world, r, 1, "gcn", C, pieces) for r in range(world)]
        for half in ("fwd", "bwd"):
            hs = [getattr(p, half) for p in plans]
            # ranks of one row group hold the same rows; over the R groups every rewritten edge appears once
            assert sum(hs[r * C].nnz for r in range(world // C)) == plans[0].nnz_total
            for p, h in enumerate(hs):
                assert h.members == list(range((p // C) * C, (p // C + 1) * C))
                assert h.n_group == b[(p // C + 1) * C] - b[(p // C) * C]
                assert sorted(torch.unique(h.agg).tolist()) == sorted(set(h.agg.tolist())) and int(h.agg.max()) < h.n_group
                assert h.piece_ptr[-1] == h.n_group and len(h.piece_ptr) == pieces + 1
                for k in range(pieces):
                    assert sum(h.piece_counts[k]) == h.piece_ptr[k + 1] - h.piece_ptr[k]
                    assert all(h.piece_counts[k][q] == 0 for q in range(world) if q not in h.members)
                # what the members send me in piece k is what I expect to receive
                for k in range(pieces):
                    a, e = h.my_piece[k]
                    for q in h.members:
                        assert hs[q].piece_counts[k][p] == e - a


@pytest.mark.parametrize("world", [2, 3, 4, 6])
def test_plans_from_edge_list_slices_are_the_plans_from_the_whole_list(world, tmp_path):
    """DistGraph(plan_from_slices=True) / RGBX_PLAN_FROM_SLICES=1: every rank buckets its 1/P of the edge list by owner and one
    all-to-all of edge records per direction hands every rank (or row group) its edges in global order; halo plans and R x C
    grid plans built from them equal the ones built from the whole list in every tensor and count, and the propagate gives
    the same rows and gradients (worlds 2-6: uneven node ranges, empty buckets, a graph with as many nodes as ranks)."""
    mp.spawn(W.plan_slices_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    parts = [torch.load(os.path.join(tmp_path, f"slices_{r}.pt")) for r in range(world)]
    assert all(p["checked"] > 500 for p in parts)


@pytest.mark.parametrize("world,exchange", [(2, "halo"), (3, "reshard"), (3, "auto"), (4, "2x2"), (6, "2x3")])
def test_distributed_propagate_matches_single_process(world, exchange, tmp_path):
    mp.spawn(W.propagate_worker, args=(world, _free_port(), str(tmp_path), exchange), nprocs=world, join=True)
    ei, x, _, _ = W.make_problem()
    n = x.size(0)
    go = torch.randn(n, x.size(1), generator=torch.Generator().manual_seed(5))
    parts = [torch.load(os.path.join(tmp_path, f"prop_{r}.pt")) for r in range(world)]
    schemes = parts[0]["schemes"]
    assert all(p["schemes"] == schemes for p in parts)  # every rank takes the same branch
    if exchange in ("halo", "reshard"):
        assert set(schemes) == {exchange}  # feature width 12 is divisible by 2 and 3
    elif "x" in exchange:
        assert set(schemes) == {"grid" + exchange}
    else:  # the cost model's pick: whatever it is, one of the known schemes (and the same on every rank)
        assert set(schemes) <= {"halo", "reshard"}
    for mode, kind in ((1, "gcn"), (2, "mean"), (0, "mean"), (0, "sum")):
        rei, _ = O.rewrite_edges(ei, n, mode)
        xr = x.clone().requires_grad_(True)
        if kind == "gcn":
            _, w = O.gcn_norm(ei, None, n)
            want = O.propagate(rei, xr, n, w, "add")
        else:
            want = O.propagate(rei, xr, n, None, "add" if kind == "sum" else "mean")
        want.backward(go)
        out = torch.cat([p[f"{mode}_{kind}"][0] for p in parts])
        grad = torch.cat([p[f"{mode}_{kind}"][1] for p in parts])
        assert torch.allclose(out, want.detach(), atol=1e-5), (mode, kind)
        assert torch.allclose(grad, xr.grad, atol=1e-5), (mode, kind)
    xr = x.clone().requires_grad_(True)
    want = O.appnp(xr, ei, 4, 0.15)
    want.backward(go)
    assert torch.allclose(torch.cat([p["appnp"][0] for p in parts]), want.detach(), atol=1e-5)
    assert torch.allclose(torch.cat([p["appnp"][1] for p in parts]), xr.grad, atol=1e-5)


def _single_process_reference(model_name):
    """The same three epochs, one process, whole graph, with the oracle's forward under autograd."""
    from rgb_experiment_amd import models as M
    ei, x, y, masks = W.make_problem()
    torch.manual_seed(14530529)
    model = W.build_model(M, model_name, x.size(1), int(y.max()) + 1)
    params = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()
              if "lin_dst" not in k}
    trainable = [k for k, _ in model.named_parameters()]
    opt = torch.optim.Adam([params[k] for k in trainable], lr=0.01)

    def fwd(training):
        if model_name.endswith("_grid"):
            f, layers = {"gcn_grid": (O.gcn_forward, 2), "gcn3_grid": (O.gcn_forward, 3),
                         "graphsage_grid": (O.graphsage_forward, 2), "graphsage2_grid": (O.graphsage2_forward, 3)}[model_name]
            return f(params, x, ei, layers, training)
        if model_name.endswith("_wide"):
            f = {"gcn_wide": O.gcn_forward, "graphsage_wide": O.graphsage_forward,
                 "graphsage2_wide": O.graphsage2_forward}[model_name]
            return f(params, x, ei, 2, training)
        if model_name == "gcn":
            return O.gcn_forward(params, x, ei, 3, training)
        if model_name == "graphsage":
            return O.graphsage_forward(params, x, ei, 2, training)
        if model_name == "graphsage2":
            return O.graphsage2_forward(params, x, ei, 2, training)
        if model_name == "gat":
            return O.gat_forward(params, x, ei, 2, 3, training)
        return O.appnp_stack_forward(params, x, ei, 4, 0.1, training)

    hist = []
    for _ in range(3):
        opt.zero_grad()
        bn_inputs = {}
        out = fwd(True)["out"]
        loss = torch.nn.functional.nll_loss(out[masks[0]], y[masks[0]])
        loss.backward()
        opt.step()
        with torch.no_grad():
            ev = fwd(False)["out"]
            vl = torch.nn.functional.nll_loss(ev[masks[1]], y[masks[1]]).item()
            sl = torch.nn.functional.nll_loss(ev[masks[2]], y[masks[2]]).item()
        hist.append((loss.item(), vl, sl))
    return hist, params


@pytest.mark.parametrize("model_name", ["gcn"])  # (appnpstack: tests/test_gpu_dist.py, on the real kernels)
def test_experiment_runs_as_one_of_several_ranks(model_name, tmp_path):
    """experiment() under WORLD_SIZE > 1 takes the node-partitioned route (dist/experiment.py): every rank returns the
    same metrics and trained weights, and 2 ranks train like 3 ranks (the partition changes summation orders only)."""
    runs = {}
    for world in (2, 3):
        mp.spawn(W.experiment_worker, args=(world, _free_port(), str(tmp_path), model_name), nprocs=world, join=True)
        runs[world] = [torch.load(os.path.join(tmp_path, f"exp_{model_name}_{world}_{r}.pt")) for r in range(world)]
        first = runs[world][0]
        assert first["distributed"]["world"] == world and len(first["history"]["val_loss"]) == 6
        for other in runs[world][1:]:
            assert other["metrics"] == first["metrics"] and other["history"]["val_loss"] == first["history"]["val_loss"]
            for k, v in first["state"].items():
                assert torch.equal(v, other["state"][k]), k
    a, b = runs[2][0], runs[3][0]
    assert np.allclose(a["history"]["train_loss"], b["history"]["train_loss"], rtol=0, atol=2e-5)
    assert abs(a["metrics"]["ACC"] - b["metrics"]["ACC"]) <= 0.11  # 19 test nodes: two of them may flip


def test_experiment_takes_the_task_split_where_it_pays(tmp_path):
    """experiment() with an APPNP stack of 40 classes on 4 ranks: 4 column slices of 10 floats would fall below the
    128-byte line, two groups of 2 ranks do not (dist.tasksplit.pays) — ranks 0-1 train, ranks 2-3 evaluate, every rank
    returns the same history, metrics (each test row counted once) and weights; same training as 3 plain ranks."""
    runs = {}
    for world in (4, 3):
        mp.spawn(W.experiment_worker, args=(world, _free_port(), str(tmp_path), "appnpstack", False, 40), nprocs=world,
                 join=True)
        runs[world] = [torch.load(os.path.join(tmp_path, f"exp_appnpstack_{world}_{r}.pt")) for r in range(world)]
        first = runs[world][0]
        for other in runs[world][1:]:
            assert other["metrics"] == first["metrics"]
            for key in ("train_loss", "val_loss", "val_acc", "test_loss", "test_acc"):  # (train_acc: nan on this route)
                assert other["history"][key] == first["history"][key], key
            for k, v in first["state"].items():
                assert torch.equal(v, other["state"][k]), k
    a, b = runs[4][0], runs[3][0]
    assert a["distributed"]["test_rows"] == b["distributed"]["test_rows"]  # both groups hold every row: one reports
    assert [p["distributed"]["task_split_role"] for p in runs[4]] == ["train", "train", "eval", "eval"]
    assert runs[3][0]["distributed"]["task_split_role"] is None
    assert np.allclose(a["history"]["train_loss"], b["history"]["train_loss"], rtol=0, atol=2e-5)
    assert np.allclose(a["history"]["val_loss"], b["history"]["val_loss"], rtol=0, atol=3e-2)
    assert abs(a["metrics"]["ACC"] - b["metrics"]["ACC"]) <= 0.11


@pytest.mark.parametrize("model_name,with_resident,without", [("gcn", 8, 11), ("gat", 4, 8)])
def test_resident_input_features_remove_the_first_layer_exchange(model_name, with_resident, without, tmp_path):
    """The boundary rows of the static feature matrix are fetched once (DistGraph.pin_resident): a steady-state
    epoch (1 train forward + backward, 2 eval forwards) then exchanges only for the layers whose input is an
    activation, and the numbers do not move. gcn (3 layers): 3x3 forward + 2 backward exchanges -> 3x2 + 2;
    graphsage2 (2 layers): 3x2 + 1 -> 3x1 + 1; gat (2 layers): 3x2 + 2 -> 3x1 + 1 (the halo rows of h = xW
    are recomputed from the resident x, so layer 1 needs no reverse exchange either)."""
    res = {}
    for resident in (True, False):
        mp.spawn(W.exchange_count_worker, args=(2, _free_port(), str(tmp_path), model_name, resident), nprocs=2,
                 join=True)
        res[resident] = [torch.load(os.path.join(tmp_path, f"cnt_{model_name}_{int(resident)}_{r}.pt"))
                         for r in range(2)]
    assert res[True][0]["exchanges"] == res[True][1]["exchanges"] == with_resident
    assert res[False][0]["exchanges"] == res[False][1]["exchanges"] == without
    for a, b in zip(res[True][0]["hist"], res[False][0]["hist"]):
        assert abs(a[0] - b[0]) < 1e-5, (a, b)  # train loss
        # eval losses see the pre-BatchNorm biases, whose zero true gradient Adam turns into +-lr noise that
        # depends on the summation order (see test_dist_runner_training_matches_single_process)
        assert abs(a[1] - b[1]) < 5e-3 and abs(a[3] - b[3]) < 5e-3, (a, b)


def test_interleaved_eval_forwards_equal_sequential_ones(tmp_path):
    """DistRunner._interleaved_evals (val and test forward issued by two host threads that take turns at their
    exchange waits) gives the numbers of two forwards in a row, and the same on every rank."""
    res = {}
    for interleave in (True, False):
        mp.spawn(W.runner_worker, args=(2, _free_port(), str(tmp_path), "gcn", "reshard", interleave), nprocs=2,
                 join=True)
        res[interleave] = [torch.load(os.path.join(tmp_path, f"run_gcn_{r}.pt")) for r in range(2)]
    assert res[True][0]["hist"] == res[False][0]["hist"] == res[True][1]["hist"]
    for k, v in res[True][0]["state"].items():
        assert torch.equal(v, res[False][0]["state"][k]), k


@pytest.mark.timeout(300)
def test_an_error_inside_an_interleaved_eval_forward_ends_the_job(tmp_path):
    """One rank's exception on an eval thread used to be carried to a join while its peers waited for ever in the next
    all-to-all. Now the failing rank's process ends at once (Comm.abort, status 70) and the spawner ends the rest."""
    with pytest.raises(Exception) as info:
        mp.spawn(W.failing_eval_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert "70" in str(info.value) or "exit" in str(info.value).lower(), info.value
    assert os.path.exists(os.path.join(tmp_path, "first_epoch_done_0"))
    assert os.path.exists(os.path.join(tmp_path, "first_epoch_done_1"))
    assert not os.path.exists(os.path.join(tmp_path, "second_epoch_done_1"))


@pytest.mark.parametrize("world,exchange", [(2, "reshard"), (3, "reshard"), (4, "2x2")])
def test_pipelined_reshard_equals_the_single_exchange(world, exchange, tmp_path):
    """DistGraph._aggregate_and_return (the row group's rows in piece-major order, one SpMM per piece, each piece's
    all-to-all in flight while the next is aggregated) returns exactly the rows of the one-shot exchange, forward
    and backward, also with ragged pieces (103 nodes, 3 pieces, 3 ranks; f = 12 columns, 4 per rank)."""
    mp.spawn(W.reshard_chunk_worker, args=(world, _free_port(), str(tmp_path), exchange), nprocs=world, join=True)
    for r in range(world):
        outs = torch.load(os.path.join(tmp_path, f"chunks_{r}.pt"))
        for chunks in (3, 4):
            assert torch.equal(outs[chunks][0], outs[1][0]) and torch.equal(outs[chunks][1], outs[1][1]), (r, chunks)


@pytest.mark.parametrize("world,sabotage", [(2, False), (3, False), (4, False), (4, True)])
def test_communicator_self_test_before_the_fused_schedule(world, sabotage, tmp_path):
    """Comm.self_test_views, run by DistRunner before the fused schedule's first epoch: aliased / empty / row-range views
    in one exchange, then two piece-wise view exchanges plus an async all-reduce in flight together and waited for out of
    issue order, each with a known answer; every rank gets the same verdict, and a rank that receives wrong rows in the
    second piece turns it into False for ALL ranks (the runner then stays on the single-buffer exchanges)."""
    mp.spawn(W.comm_selftest_worker, args=(world, _free_port(), str(tmp_path), sabotage), nprocs=world, join=True)
    parts = [torch.load(os.path.join(tmp_path, f"selftest_{r}.pt")) for r in range(world)]
    assert [p["ok"] for p in parts] == [not sabotage] * world
    assert all(p["exchanges"] == 0 for p in parts)  # the self-test leaves the traffic counters clean


def test_dist_batchnorm_matches_full_batch_bn():
    """world = 1 (no process group): DistBatchNorm1d == nn.BatchNorm1d incl. running stats and grads."""
    from rgb_experiment_amd.dist import Comm, DistBatchNorm1d
    torch.manual_seed(0)
    ref = torch.nn.BatchNorm1d(6)
    with torch.no_grad():
        ref.weight.uniform_(0.5, 2)
        ref.bias.uniform_(-1, 1)
    mine = DistBatchNorm1d.convert(torch.nn.Sequential(copy.deepcopy(ref)), Comm())[0]
    assert isinstance(mine, DistBatchNorm1d) and list(mine.state_dict()) == list(ref.state_dict())
    x = torch.randn(40, 6) * 3 + 1
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb = ref(xa), mine(xb)
    go = torch.randn(40, 6)
    ya.backward(go)
    yb.backward(go)
    assert torch.allclose(ya, yb, atol=1e-5) and torch.allclose(xa.grad, xb.grad, atol=1e-5)
    assert torch.allclose(ref.weight.grad, mine.weight.grad, atol=1e-5)
    assert torch.allclose(ref.bias.grad, mine.bias.grad, atol=1e-5)
    assert torch.allclose(ref.running_mean, mine.running_mean, atol=1e-6)
    assert torch.allclose(ref.running_var, mine.running_var, atol=1e-5)
    ref.eval(), mine.eval()
    assert torch.allclose(ref(x), mine(x), atol=1e-6)


@pytest.mark.parametrize("idents,shared", [(["box|0:5:0", "box|0:15:0", "box|0:25:0"], False),
                                           (["box|0:5:0", "box|0:5:0", "box|0:25:0"], True)])
def test_device_identities_through_the_rendezvous_store(idents, shared, tmp_path):
    """dist/sharing.py: the ranks publish what device they opened through the env:// rendezvous store before any communicator
    exists, every rank reads every identity, and the process group is then made from that same store (the route init_rccl
    takes; here on gloo). One visible GPU per rank with DIFFERENT identities is not 'sharing'."""
    mp.spawn(W.identity_exchange_worker, args=(3, _free_port(), str(tmp_path), idents), nprocs=3, join=True)
    for r in range(3):
        p = torch.load(os.path.join(tmp_path, f"ident_{r}.pt"))
        assert p["got"] == idents and p["shared"] is shared and p["sum"] == 6.0 and (p["rank"], p["world"]) == (r, 3)


@pytest.mark.parametrize("world", [2, 3, 4, 6, 8])
def test_partition_plans_schemes_and_runner_on_random_problems(world, tmp_path):
    """Two fuzzes in one set of rank processes per world size (W.dist_fuzz_worker). (1) The distributed propagate (every loops
    mode / kind) and K-step APPNP, forward and backward, on random problems — node counts from one row per rank, edge lists
    from empty to hub-heavy, widths that do and do not divide by the world size or the grid's column count, every exchange
    scheme and piece count — against the oracle in float64. (2) DistRunner's two epochs (fused per-rank schedule or modules,
    every scheme incl. replicate, random model family / widths / depth) against single-process oracle training: train losses
    (the second sees the first update), and the last epoch's eval numbers against the oracle's forward on the run's own
    weights. 10 + 5 pinned seeds per world size; RGBX_DIST_FUZZ_SEEDS=<count> [RGBX_DIST_FUZZ_FIRST=<offset>] soaks more
    (round 4: 5,000 + 1,500 cases, no failure after the ZeroDivisionError of an empty eval mask was turned into nan)."""
    count = int(os.environ.get("RGBX_DIST_FUZZ_SEEDS", "0"))
    first = int(os.environ.get("RGBX_DIST_FUZZ_FIRST", "0")) + 1000 * world
    prop, run = (range(first, first + count),) * 2 if count else (range(first, first + 10), range(first, first + 5))
    mp.spawn(W.dist_fuzz_worker, args=(world, _free_port(), str(tmp_path), list(prop), list(run)), nprocs=world, join=True)
    bad = [b for name in ("propfuzz", "runfuzz") for r in range(world) for b in torch.load(os.path.join(tmp_path, f"{name}_{r}.pt"))]
    assert not bad, bad[:6]
