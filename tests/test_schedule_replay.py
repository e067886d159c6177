"""bench.replay_schedule: the emulated rank's recorded schedule against a FIFO link model."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))


class _Ev:
    """Stand-in for a pair of HIP events `ms` apart."""

    def __init__(self, ms=None):
        self.ms = ms

    def elapsed_time(self, end):
        return end.ms


def _launch(kind, ms):
    return (kind, _Ev(), _Ev(ms))


def _replay(trace, steps=1, **kw):
    import bench
    r = bench.replay_schedule(trace, steps, gbs=(1.0,), **kw)  # 1 GB/s: 1e6 bytes take 1 ms
    return r, r["by_link_rate"]["1 GB/s per link and direction"]


def test_an_exchange_hidden_behind_later_launches():
    # 1 ms launch, exchange of 0.5 ms issued, 0.3 ms launch, wait: the stream stands still for the missing 0.2 ms
    trace = [_launch("a", 1.0), ("@issue", 1, "x", 500_000), _launch("b", 0.3), ("@wait", 1), _launch("c", 1.0)]
    r, at = _replay(trace)
    assert abs(at["exposed_ms_per_epoch"] - 0.2) < 1e-9 and abs(at["replayed_ms_per_epoch"] - 2.5) < 1e-9
    assert abs(r["timed_launch_ms_per_epoch"] - 2.3) < 1e-9 and r["exchanges_per_epoch"] == 1
    assert at["stalls_by_exchange_ms"] == {"x": 0.2}
    # enough work behind it: nothing shows
    trace[2] = _launch("b", 0.7)
    assert _replay(trace)[1]["exposed_ms_per_epoch"] == 0.0


def test_exchanges_share_one_fifo_link_and_start_after_their_producer():
    # two exchanges issued back to back after a 1 ms producer: the second one queues behind the first
    trace = [_launch("p", 1.0), ("@issue", 1, "first", 400_000), ("@issue", 2, "second", 400_000),
             _launch("q", 0.5), ("@wait", 2), ("@wait", 1)]
    _, at = _replay(trace)
    # link: 1.0 -> 1.4 -> 1.8; compute reaches the wait at 1.5: stands still until 1.8, the first one is long done
    assert abs(at["exposed_ms_per_epoch"] - 0.3) < 1e-9 and at["stalls_by_exchange_ms"] == {"second": 0.3}
    # an exchange cannot start before the launches enqueued ahead of its issue have finished
    trace = [("@issue", 1, "early", 100_000), _launch("p", 1.0), ("@issue", 2, "late", 100_000), ("@wait", 2)]
    _, at = _replay(trace)
    assert abs(at["exposed_ms_per_epoch"] - 0.1) < 1e-9 and list(at["stalls_by_exchange_ms"]) == ["late"]


def test_latency_per_exchange_and_per_epoch_averages():
    step = [_launch("a", 1.0), ("@issue", 1, "x", 200_000), ("@wait", 1)]
    trace = step + [_launch("a", 1.0), ("@issue", 2, "x", 200_000), ("@wait", 2)]
    _, at = _replay(trace, steps=2)
    assert abs(at["exposed_ms_per_epoch"] - 0.2) < 1e-9
    _, at = _replay(trace, steps=2, latency_us=50.0)
    assert abs(at["exposed_ms_per_epoch"] - 0.25) < 1e-9


def test_where_the_epoch_is_split_by_task():
    """dist.tasksplit.pays: every model at two ranks (one link between two GPUs: no partition scheme pays), APPNP stacks
    on an even number of >= 4 ranks whose P column slices fall below 32 floats while P / 2 slices keep 16; nothing else."""
    from rgb_experiment_amd import models as M
    from rgb_experiment_amd.dist import tasksplit
    appnp = M.APPNPStack(hidden_unit=16, input_dim=12, output_dim=128, K=10, alpha=0.1, dropout_rate=0.5)
    gcn = M.GCN(num_layers=2, hidden_unit=128, input_dim=128, output_dim=128, dropout_rate=0.5)
    assert tasksplit.pays(gcn, 2) and tasksplit.pays(appnp, 2)
    assert tasksplit.pays(appnp, 8) and tasksplit.pays(appnp, 6)          # 16- / 21-float slices; halves keep 32 / 42
    assert not tasksplit.pays(appnp, 4) and not tasksplit.pays(appnp, 7)  # 32-float slices already; odd world
    assert not tasksplit.pays(gcn, 8) and not tasksplit.pays(gcn, 4)
    small = M.APPNPStack(hidden_unit=16, input_dim=12, output_dim=40, K=10, alpha=0.1, dropout_rate=0.5)
    assert tasksplit.pays(small, 4) and not tasksplit.pays(small, 8)      # 40 classes: 10-float slices; 8 ranks: halves keep 10 only
    assert tasksplit.pays(appnp, 8, width=128) and not tasksplit.pays(appnp, 8, width=512)


def test_task_split_resolution_and_memory_guard(monkeypatch):
    """experiment(task_split=...) / RGBX_TASK_SPLIT: 'off' always partitions, 'on' always splits (even worlds), 'auto' =
    pays() — and on TWO ranks, where the split keeps the whole graph on both GPUs, only when one GPU can hold it."""
    from rgb_experiment_amd import models as M
    from rgb_experiment_amd.dist import tasksplit
    gcn = M.GCN(num_layers=2, hidden_unit=128, input_dim=128, output_dim=128, dropout_rate=0.5)
    cpu = torch.device("cpu")
    args = (2_000_000, 60_000_000, [128], cpu)
    assert tasksplit.resolve("auto", gcn, 2, *args) and tasksplit.resolve("on", gcn, 2, *args)
    assert not tasksplit.resolve("off", gcn, 2, *args) and not tasksplit.resolve("auto", gcn, 8, *args)
    assert not tasksplit.resolve("on", gcn, 3, *args)  # odd world: no two equal groups
    monkeypatch.setenv("RGBX_TASK_SPLIT", "off")
    assert not tasksplit.resolve("auto", gcn, 2, *args) and tasksplit.resolve("on", gcn, 2, *args)
    monkeypatch.delenv("RGBX_TASK_SPLIT")
    monkeypatch.setattr(tasksplit, "whole_graph_fits", lambda *a, **k: False)  # a graph one GPU cannot hold
    assert not tasksplit.resolve("auto", gcn, 2, *args) and tasksplit.resolve("on", gcn, 2, *args)
    with pytest.raises(ValueError):
        tasksplit.resolve("maybe", gcn, 2, *args)
    # the footprint estimate against the measured peak of the benchmark epoch (9.7 GB at L)
    assert 8e9 < tasksplit.whole_graph_bytes(2_000_000, 60_000_000, [128]) < 12e9


def test_task_split_refuses_a_grid_that_does_not_factor_its_groups():
    """TaskSplitRunner checks an explicit RxC exchange against the GROUP size in its constructor, before any rank creates a
    group or waits for another: 4 ranks are 2 groups of 2, "2x2" cannot be laid over a group."""
    import pytest
    import torch

    from rgb_experiment_amd.dist import TaskSplitRunner
    with pytest.raises(ValueError, match="task-split group"):
        TaskSplitRunner(None, None, None, None, None, 0, 4, torch.device("cpu"), exchange="2x2")


def test_dist_runner_refuses_a_grid_that_does_not_factor_the_world():
    import pytest
    import torch

    from rgb_experiment_amd.dist import DistRunner
    from rgb_experiment_amd.dist.comm import EmulatedComm
    with pytest.raises(ValueError, match="does not factor the world size 4"):
        DistRunner(None, None, torch.zeros(8, 4), torch.zeros(8), [], 0, 4, torch.device("cpu"), comm=EmulatedComm(4, 0),
                   exchange="2x4")
