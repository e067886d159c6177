"""The per-rank schedules on gloo: DistRunner's module route, the fused per-rank schedule, the step computed ahead, the shared eval
forward, the epoch split by task — against the single-process run. (Second half of tests/test_dist_gloo.py: a file of its own
so that `-n` workers of pytest-xdist, which distribute by file here, share the multi-rank cases.)"""
import os

import pytest
import torch
import torch.multiprocessing as mp

import _dist_worker as W
from test_dist_gloo import _free_port, _single_process_reference



@pytest.mark.parametrize("model_name,world,exchange", [("gcn", 3, "halo"), ("graphsage", 2, "halo"),
                                                        ("graphsage2", 2, "halo"), ("gat", 2, "halo"),
                                                        ("appnpstack", 2, "reshard"), ("gcn", 4, "2x2"),
                                                        ("appnpstack", 4, "2x2"),
                                                        # first layer replicated, second on a rectangular CSR, no
                                                        # activation exchange (dist.ReplicaGraph); "gcn" has a third
                                                        # layer, which exchanges as usual
                                                        ("gcn_wide", 2, "replicate"), ("graphsage_wide", 3, "replicate"),
                                                        ("gcn", 3, "replicate")])
def test_dist_runner_training_matches_single_process(model_name, world, exchange, tmp_path):
    """Train-mode BatchNorm uses batch statistics in the oracle and reduced statistics in the runner, so
    train losses and trained WEIGHTS must agree; eval losses use running statistics, which the
    oracle's functional BN does not update, so they are compared in a separate running-stat-free way:
    every rank's trained parameters are identical and equal to the single-process ones."""
    mp.spawn(W.runner_worker, args=(world, _free_port(), str(tmp_path), model_name, exchange, True, True),
             nprocs=world, join=True)
    parts = [torch.load(os.path.join(tmp_path, f"run_{model_name}_{r}.pt")) for r in range(world)]
    hist, params = _single_process_reference(model_name)
    for r in range(1, world):  # replicated parameters stay bit-identical across ranks
        for k, v in parts[0]["state"].items():
            assert torch.equal(v, parts[r]["state"][k]), k
        assert parts[0]["hist"] == parts[r]["hist"]
    for step in range(3):
        assert abs(parts[0]["hist"][step][0] - hist[step][0]) < 2e-5, (step, parts[0]["hist"][step], hist[step])
    # A bias added right before a BatchNorm has an exactly-zero true gradient (BN removes constant
    # shifts); Adam turns its rounding noise into +-lr steps, so those entries are not comparable.
    last = {"gcn": "convs.2.", "graphsage": "convs.1.", "graphsage2": "convs.1.", "appnpstack": "lin2.",
            "gat": "convs.1.", "gcn_wide": "convs.1.", "graphsage_wide": "convs.1.",
            "graphsage2_wide": "convs.1."}[model_name]
    for k, v in params.items():
        pre_bn_bias = k.endswith("bias") and not k.startswith(("bns.", "bn.", last))
        if v.is_floating_point() and "running" not in k and not pre_bn_bias:
            assert torch.allclose(parts[0]["state"][k], v.detach(), atol=2e-5), k
    assert parts[0]["lo"] == 0 and parts[-1]["hi"] == 97


@pytest.mark.parametrize("model_name,world,exchange,fused,split", [
    ("gcn_grid", 2, "reshard", True, False), ("graphsage_grid", 4, "2x2", True, False), ("gcn3_grid", 3, "reshard", True, False),
    ("gcn", 3, "halo", False, False), ("appnpstack", 2, "reshard", False, False), ("appnpstack", 4, "reshard", False, True)])
def test_shared_eval_forward_on_several_ranks(model_name, world, exchange, fused, split, tmp_path):
    """share_eval_forward in DistRunner / GridStack / TaskSplitRunner (reference loop: itexperiments.py:464-473 runs two
    identical eval forwards): the five numbers of every epoch and the trained state equal those of the two-forward epoch
    bit for bit, with fewer exchanges per epoch (one eval forward's worth)."""
    mp.spawn(W.shared_eval_worker, args=(world, _free_port(), str(tmp_path), model_name, exchange, fused, split),
             nprocs=world, join=True)
    parts = [torch.load(os.path.join(tmp_path, f"shared_{model_name}_{r}.pt")) for r in range(world)]
    for p in parts:
        two, one = p[False], p[True]
        assert one["hist"] == two["hist"] == parts[0][False]["hist"], (one["hist"], two["hist"])
        assert one["engine"] == two["engine"] == (fused and not split)
        for k, v in two["state"].items():
            assert torch.equal(v, one["state"][k]), k
    # exchanges: a training rank of a task split makes no eval exchange at all; everybody else saves one eval forward's
    for p in parts:
        if p[True]["role"] == "train":
            assert p[True]["exchanges"] == p[False]["exchanges"]
        else:
            assert p[True]["exchanges"] < p[False]["exchanges"], (p[True]["exchanges"], p[False]["exchanges"])
            assert p[True]["bytes"] < p[False]["bytes"]


@pytest.mark.parametrize("model_name,world,exchange,stop_early", [("appnpstack", 4, "reshard", True),
                                                                   ("gcn", 2, "halo", True),
                                                                   # BASELINE configs[4] as the 8-GPU tier runs it: 4 + 4
                                                                   ("appnpstack", 8, "reshard", False)])
def test_epoch_split_by_task_over_two_groups(model_name, world, exchange, stop_early, tmp_path):
    """dist.TaskSplitRunner: ranks [0, P/2) run the training steps, ranks [P/2, P) the val and test forwards, each group
    with the whole graph partitioned over its ranks; the training group computes step t + 1 ahead while the eval group
    evaluates the model step t left. Every rank reports the same five numbers per epoch; train losses and trained
    weights equal single-process oracle training; the eval group's model IS the training group's (state_dict bit for
    bit, also after a step computed ahead was dropped); eval losses equal those of a plain DistRunner run."""
    mp.spawn(W.tasksplit_worker, args=(world, _free_port(), str(tmp_path), model_name, exchange, stop_early),
             nprocs=world, join=True)
    parts = [torch.load(os.path.join(tmp_path, f"split_{model_name}_{r}.pt")) for r in range(world)]
    assert [p["role"] for p in parts] == ["train"] * (world // 2) + ["eval"] * (world // 2)
    assert [(p["lo"], p["hi"]) for p in parts[:world // 2]] == [(p["lo"], p["hi"]) for p in parts[world // 2:]]
    for p in parts[1:]:
        assert p["hist"] == parts[0]["hist"]
        for k, v in parts[0]["state"].items():
            assert torch.equal(v, p["state"][k]), (p["role"], k)
    hist, params = _single_process_reference(model_name)
    for step in range(3):
        assert abs(parts[0]["hist"][step][0] - hist[step][0]) < 2e-5, (step, parts[0]["hist"][step], hist[step])
    last = {"gcn": "convs.2.", "appnpstack": "lin2.", "graphsage2": "convs.1."}[model_name]
    for k, v in params.items():
        pre_bn_bias = k.endswith("bias") and not k.startswith(("bns.", "bn.", last))
        if v.is_floating_point() and "running" not in k and not pre_bn_bias:
            assert torch.allclose(parts[0]["state"][k], v.detach(), atol=2e-5), k
    # the eval statistics against a plain DistRunner over world / 2 ranks (the same partition as a group's)
    if world == 2:  # (groups of one rank: nothing to compare a partition with)
        return
    mp.spawn(W.runner_worker, args=(world // 2, _free_port(), str(tmp_path), model_name, exchange, False, False, False),
             nprocs=world // 2, join=True)
    ref = torch.load(os.path.join(tmp_path, f"run_{model_name}_0.pt"))
    for a, b in zip(parts[0]["hist"], ref["hist"]):
        assert abs(a[0] - b[0]) < 2e-5 and abs(a[1] - b[1]) < 3e-2 and abs(a[3] - b[3]) < 3e-2, (a, b)
        assert abs(a[2] - b[2]) < 0.05 and abs(a[4] - b[4]) < 0.05


@pytest.mark.parametrize("model_name,world,exchange,pieces,also_modules", [
    ("gcn_grid", 2, "reshard", 1, True), ("gcn3_grid", 4, "2x2", 3, False), ("graphsage_grid", 2, "reshard", 2, False),
    ("graphsage_grid", 6, "2x3", 1, False), ("graphsage2_grid", 4, "2x2", 4, True), ("gcn3_grid", 6, "3x2", 2, False),
    # the world size the 8-GPU tier runs, the scheme the cost model picks there for the benchmark (2 row groups x 4 slices)
    ("gcn_grid", 8, "2x4", 2, False), ("graphsage_grid", 8, "2x4", 4, False)])
def test_fused_grid_schedule_matches_single_process(model_name, world, exchange, pieces, also_modules, tmp_path):
    """dist/stack.py GridStack (layer outputs written blocked into the send buffers, BatchNorm / transform / loss in the
    return stage, manual backward, view exchanges) trains exactly like one process running the oracle under autograd:
    train losses of three epochs, every trained parameter, and the eval losses against the module path of the same run."""
    mp.spawn(W.runner_worker, args=(world, _free_port(), str(tmp_path), model_name, exchange, True, True, True, pieces),
             nprocs=world, join=True)
    parts = [torch.load(os.path.join(tmp_path, f"run_{model_name}_{r}.pt")) for r in range(world)]
    assert all(p["engine"] for p in parts), "the fused schedule was not taken"
    hist, params = _single_process_reference(model_name)
    for r in range(1, world):
        for k, v in parts[0]["state"].items():
            assert torch.equal(v, parts[r]["state"][k]), k
        assert parts[0]["hist"] == parts[r]["hist"]
    for step in range(3):
        assert abs(parts[0]["hist"][step][0] - hist[step][0]) < 2e-5, (step, parts[0]["hist"][step], hist[step])
    n_layers = 3 if model_name in ("gcn3_grid", "graphsage2_grid") else 2
    last = f"convs.{n_layers - 1}."
    for k, v in params.items():
        pre_bn_bias = k.endswith("bias") and not k.startswith(("bns.", last))
        # a mean aggregation (weights sum to 1 on every row: each of the 97 nodes has in-edges) carries BatchNorm 0's
        # shift as a constant row into the next BatchNorm, which removes it: zero true gradient there as well
        pre_bn_bias = pre_bn_bias or (model_name == "graphsage2_grid" and k == "bns.0.bias")
        if v.is_floating_point() and "running" not in k and not pre_bn_bias:
            assert torch.allclose(parts[0]["state"][k], v.detach(), atol=2e-5), k
    # eval forwards of the SAME weights, fused schedule vs modules, on every rank: NLL sums to rounding, hits equal
    for p in parts:
        for eng, mod_stats in zip(*p["eval_both"]):
            assert abs(eng[0].item() - mod_stats[0].item()) < 1e-4 * max(1.0, abs(mod_stats[0].item())), (eng, mod_stats)
            assert eng[1].item() == mod_stats[1].item()
    if not also_modules:
        return
    # the same run through the modules (fused=False): identical schedule of collectives aside, the numbers agree
    mp.spawn(W.runner_worker, args=(world, _free_port(), str(tmp_path), model_name, exchange, False, True, False, pieces),
             nprocs=world, join=True)
    mod = torch.load(os.path.join(tmp_path, f"run_{model_name}_0.pt"))
    assert not mod["engine"]
    for a, b in zip(parts[0]["hist"], mod["hist"]):
        # (eval losses of two separately trained runs differ by Adam's +-lr noise on the pre-BatchNorm biases, see
        # test_dist_runner_training_matches_single_process; the eval forward is compared on the same weights below)
        assert abs(a[0] - b[0]) < 2e-5 and abs(a[1] - b[1]) < 3e-2 and abs(a[3] - b[3]) < 3e-2, (a, b)


@pytest.mark.parametrize("model_name,world,exchange,pieces", [("gcn3_grid", 3, "reshard", 3)])  # (+ the 2 x 2 grid in -m gpu)
def test_next_training_step_computed_during_the_eval_forwards(model_name, world, exchange, pieces, tmp_path):
    """DistRunner.epoch(more=True): the eval forwards of an epoch are interleaved with the forward + backward of the
    NEXT epoch's training step (one thread; its optimizer step waits for the next call). Same kernels on the same
    operands: losses, accuracies and the whole state_dict (running statistics and their counter included) equal the
    sequential schedule's bit for bit — also when the loop stops with a step computed ahead (nothing of it may show)."""
    runs = {}
    for ahead in (None, "all", "stop"):
        mp.spawn(W.runner_worker, args=(world, _free_port(), str(tmp_path), model_name, exchange, True, True, True, pieces,
                                        False, ahead), nprocs=world, join=True)
        runs[ahead] = [torch.load(os.path.join(tmp_path, f"run_{model_name}_{r}.pt")) for r in range(world)]
    base = runs[None]
    assert all(p["engine"] for p in base)
    for ahead in ("all", "stop"):
        for p, q in zip(base, runs[ahead]):
            assert p["hist"] == q["hist"], (ahead, p["hist"], q["hist"])
            for k, v in p["state"].items():
                assert torch.equal(v, q["state"][k]), (ahead, k)
            assert torch.equal(p["logits_eval"], q["logits_eval"])
            for a, b in zip(p["eval_both"][0], q["eval_both"][0]):
                assert torch.equal(a, b)


@pytest.mark.parametrize("model_name,world,exchange,pieces", [("graphsage_grid", 4, "2x2", 2)])
def test_one_rank_without_row_range_launches(model_name, world, exchange, pieces, tmp_path, monkeypatch):
    """A rank whose CSRs carry a hub-row plan launches whole row groups and layer 0 in one piece; which ranks do depends
    on the graph. The order and number of the collectives must not: every rank takes the same number of layer-0 pieces
    and the generators yield at the same places (a first build let rank-dependent yields reorder the interleaved
    collectives: wrong rows, no error). Same numbers as the run where every rank launches row ranges."""
    runs = []
    for hub in (None, "1"):
        if hub is None:
            monkeypatch.delenv("RGBX_TEST_HUB_RANK", raising=False)
        else:
            monkeypatch.setenv("RGBX_TEST_HUB_RANK", hub)
        mp.spawn(W.runner_worker, args=(world, _free_port(), str(tmp_path), model_name, exchange, True, True, True, pieces,
                                        False, "all"), nprocs=world, join=True)
        runs.append([torch.load(os.path.join(tmp_path, f"run_{model_name}_{r}.pt")) for r in range(world)])
    n_layers = 3 if model_name in ("gcn3_grid", "graphsage2_grid") else 2
    for p, q in zip(*runs):
        assert p["engine"] and q["engine"]
        for a, b in zip(p["hist"], q["hist"]):
            # (other piece counts = other summation orders: the pre-BatchNorm biases, whose true gradient is zero, take
            # Adam's +-lr steps on rounding noise and move the eval-mode losses in the third decimal — see
            # test_dist_runner_training_matches_single_process; wrong ROWS would move everything by O(1))
            assert abs(a[0] - b[0]) < 1e-5 and abs(a[1] - b[1]) < 3e-2 and abs(a[3] - b[3]) < 3e-2, (a, b)
        for k, v in p["state"].items():
            pre_bn_bias = k.endswith("bias") and not k.startswith(("bns.", f"convs.{n_layers - 1}."))
            if v.is_floating_point() and "running" not in k and not pre_bn_bias:
                assert torch.allclose(v, q["state"][k], atol=2e-5), k


@pytest.mark.parametrize("model_name,world,exchange", [("graphsage_grid", 4, "2x2")])
def test_fused_grid_schedule_with_the_kept_input_aggregate(model_name, world, exchange, tmp_path):
    """cache_input_aggregate=True on the partitioned run (opt-in): layer 0 transforms the kept aggregate of the static
    features instead of gathering it again; same losses and weights as the recomputing run."""
    runs = []
    for cache in (False, True):
        mp.spawn(W.runner_worker, args=(world, _free_port(), str(tmp_path), model_name, exchange, True, True, True, 2,
                                        cache), nprocs=world, join=True)
        runs.append(torch.load(os.path.join(tmp_path, f"run_{model_name}_0.pt")))
    a, b = runs
    assert a["engine"] and b["engine"]
    for x, y in zip(a["hist"], b["hist"]):
        assert abs(x[0] - y[0]) < 1e-6 and abs(x[1] - y[1]) < 1e-5 and abs(x[3] - y[3]) < 1e-5, (x, y)
    for k, v in a["state"].items():
        assert torch.allclose(v, b["state"][k], atol=1e-6), k
