"""bench.py as the driver types it: `python bench.py --gpus N` must start its N ranks itself (children, before any
GPU call in the parent) and print ONE JSON line. Rehearsed here on the CPU: gloo collectives, a toy workload, and
the aggregation arithmetic injected from the oracle (the product has no CPU aggregation path)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=600, extra_env=None):
    env = dict(os.environ)
    env.update(extra_env or {})
    env.update({"RGBX_DIST_BACKEND": "gloo", "RGBX_TEST_AGGREGATOR": "_dist_worker:OracleAggregator",
                "PYTHONPATH": os.path.join(ROOT, "tests") + os.pathsep + env.get("PYTHONPATH", ""),
                "OMP_NUM_THREADS": "1", "CUDA_VISIBLE_DEVICES": "", "HIP_VISIBLE_DEVICES": ""})
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=timeout)
    return proc


def test_eight_ranks_on_the_two_by_four_grid():
    """The topology `bench.py --gpus 8` takes on the benchmark graph — 2 row groups x 4 column slices, the fused
    per-rank schedule with the next step computed ahead — with all eight ranks in play (gloo, tiny graph). (That the
    numbers do not depend on the partition is tests/test_dist_gloo.py's subject, up to 6 ranks on 2 x 3 and 3 x 2.)"""
    proc = _run(["--gpus", "8", "--workload", "T", "--steps", "2", "--warmup", "1", "--exchange", "2x4"])
    assert proc.returncode == 0, proc.stderr[-3000:]
    r8 = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1])
    assert r8["ranks_seen"] == 8 and r8["scheme"] == "grid2x4" and r8["fused_schedule"] and r8["next_step_ahead"]
    assert r8["launcher"]["attempt"] == 0
    assert all(v == v and 0 < v < 10 for v in r8["final_losses"].values())


def test_two_ranks_split_the_epoch_by_task_by_default():
    """--gpus 2: one rank trains, one evaluates (dist.tasksplit.pays: two GPUs share one link, every partition scheme
    costs more than it saves)."""
    proc = _run(["--gpus", "2", "--workload", "T", "--steps", "2", "--warmup", "1"])
    assert proc.returncode == 0, proc.stderr[-3000:]
    res = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["n_gpus"] == 2 and res["ranks_seen"] == 2 and res["task_split"]["ranks_per_group"] == 1
    assert res["next_step_ahead"] and res["launcher"]["attempt"] == 0 and res["value"] > 0
    assert all(v == v and v > 0 for v in res["final_losses"].values())


def test_bench_gpus_n_with_the_epoch_split_by_task():
    """--task-split on: ranks 0-1 train (the next step computed ahead), ranks 2-3 evaluate; one line from rank 0 with the
    split named on it and all four ranks seen."""
    proc = _run(["--gpus", "4", "--workload", "T", "--model", "appnpstack", "--steps", "2", "--warmup", "1",
                 "--task-split", "on", "--exchange", "reshard"])
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 4 and res["ranks_seen"] == 4 and res["value"] > 0
    assert res["task_split"]["ranks_per_group"] == 2 and res["task_split"]["role_of_rank_0"] == "train"
    assert res["next_step_ahead"] and not res["fused_schedule"] and res["launcher"]["attempt"] == 0
    assert res["final_losses"]["val"] == res["final_losses"]["val"] and res["final_losses"]["val"] > 0  # from the eval group


@pytest.mark.parametrize("gpus,exchange,scheme", [(3, "replicate", "replicate"), (4, "2x2", "grid2x2")])  # (2 ranks: the task-split case below)
def test_bench_gpus_n_launches_its_own_ranks(gpus, exchange, scheme):
    # (--task-split off: two ranks would otherwise split the epoch by task, each on the whole graph — its own test below)
    proc = _run(["--gpus", str(gpus), "--workload", "T", "--steps", "2", "--warmup", "1", "--exchange", exchange,
                 "--task-split", "off"])
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout  # rank 0's line, once
    res = json.loads(lines[0])
    assert res["n_gpus"] == gpus and res["ranks_seen"] == gpus and res["steps"] == 2 and res["warmup"] == 1
    if scheme is None:  # the cost model's pick (all ranks the same), with the modelled seconds on the line
        scheme = res["scheme"]
        assert scheme in ("halo", "reshard", "replicate")
        assert set(res["modelled_seconds_per_propagate"]) >= {"halo", "reshard"}
        assert set(res["modelled_seconds_per_epoch_first_two_layers"]) == {"exchange", "replicate"}
    assert res["scheme"] == scheme and res["scaling"] == "strong" and res["unit"] == "edges/s"
    assert sorted(r["rank"] for r in res["per_rank"]) == list(range(gpus))
    assert all(r["scheme"] == scheme for r in res["per_rank"])
    # the replicate scheme moves no activation rows at all; every other scheme does
    assert all((r["exchange_mb_per_step"] == 0) == (scheme == "replicate") for r in res["per_rank"])
    assert res["value"] > 0 and res["ms_per_step"] > 0
    assert res["final_losses"]["train"] == res["final_losses"]["train"]  # not NaN
    # the line names the attempt that produced it (first one here) and carries the per-rank set-up times
    assert res["launcher"]["supervised"] and res["launcher"]["attempt"] == 0 and res["launcher"]["fallback"] is None
    for r in res["per_rank"]:
        assert set(r["setup_s"]) >= {"synthetic_graph_and_masks_s", "runner_and_link_probe_s",
                                     "first_epoch_plans_and_csr_build_s"}
        assert r["exposed_exchange_ms_per_step"] >= 0
    assert "link_gbs_measured" in res and "small_all_to_all_us_measured" in res  # None on gloo, measured on RCCL
    # conv stacks under a column-slice scheme run the fused per-rank schedule (dist/stack.py)
    assert res["fused_schedule"] == (scheme in ("reshard", "grid2x2"))
    assert res["next_step_ahead"] == res["fused_schedule"]  # the next training step rides beside the eval forwards
    assert res["median_ms_per_step"] > 0


@pytest.mark.parametrize("fault", ["stall:1:0:first_epoch", "raise:0:0:timed_region"])
def test_a_failing_or_stalled_rank_ends_in_a_line_from_fresh_conservative_ranks(fault):
    """A rank of the first attempt hangs (no milestone within the stall limit) or raises: the supervisors kill that
    attempt's workers and start FRESH ones with the conservative flags; the line that comes out says which attempt
    produced it and why the first one was given up."""
    proc = _run(["--gpus", "2", "--workload", "T", "--steps", "2", "--warmup", "1", "--exchange", "reshard"],
                extra_env={"RGBX_TEST_FAULT": fault, "RGBX_LAUNCH_STALL_S": "8", "RGBX_LAUNCH_DEADLINE_S": "120"})
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout
    res = json.loads(lines[0])
    la = res["launcher"]
    assert la["attempt"] == 1 and la["extra_flags"] == ["--no-fused", "--no-interleave", "--task-split", "off", "--pieces", "1", "--pieces-in", "1"]
    assert res["scheme"] == "reshard" and res["ranks_seen"] == 2 and res["value"] > 0 and not res["fused_schedule"]
    failed = la["fallback"]["failed"]
    assert len(failed) == 1 and failed[0]["attempt"] == 0
    if fault.startswith("stall:0"):
        assert "no milestone" in failed[0]["reason"]
    elif fault.startswith("raise"):
        assert "status" in failed[0]["reason"]
    else:  # rank 1 stalled: rank 0 either saw rank 1's failure file or ran into its own stall limit
        assert "peer failed" in failed[0]["reason"] or "no milestone" in failed[0]["reason"]


@pytest.mark.parametrize("rank", [1])
def test_a_hang_after_the_line_does_not_cost_the_result(rank):
    """A rank that hangs in its TEARDOWN (a process group that does not come down) is ended by its supervisor, but the
    attempt counts: its line was out, the workers had reported "done"."""
    proc = _run(["--gpus", "2", "--workload", "T", "--steps", "2", "--warmup", "1", "--exchange", "reshard"],
                extra_env={"RGBX_TEST_FAULT": f"stall:{rank}:0:teardown", "RGBX_LAUNCH_STALL_S": "6",
                           "RGBX_LAUNCH_DEADLINE_S": "120"})
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout
    res = json.loads(lines[0])
    assert res["launcher"]["attempt"] == 0 and res["launcher"]["fallback"] is None and res["value"] > 0


def test_every_attempt_failing_ends_in_a_diagnostic_line_and_a_nonzero_status():
    proc = _run(["--gpus", "2", "--workload", "T", "--steps", "1", "--warmup", "0"],
                extra_env={"RGBX_TEST_FAULT": "raise:1:*:first_epoch", "RGBX_LAUNCH_ATTEMPTS": "2",
                           "RGBX_LAUNCH_STALL_S": "8", "RGBX_LAUNCH_DEADLINE_S": "120"})
    assert proc.returncode != 0
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout
    res = json.loads(lines[0])
    assert res["value"] is None and len(res["launcher"]["failed"]) == 2


def test_the_run_budget_bounds_all_attempts_together():
    """RGBX_LAUNCH_TOTAL_S (default 540 s, inside the driver's 600 s): a first attempt that hangs is ended at what the
    budget leaves it, a further attempt that could not finish is not started, and rank 0's diagnostic line is out in time."""
    import time
    t0 = time.time()
    proc = _run(["--gpus", "2", "--workload", "T", "--steps", "1", "--warmup", "0"],
                extra_env={"RGBX_TEST_FAULT": "stall:1:*:first_epoch", "RGBX_LAUNCH_TOTAL_S": "45",
                           "RGBX_LAUNCH_MIN_ATTEMPT_S": "20", "RGBX_LAUNCH_STALL_S": "300", "RGBX_LAUNCH_DEADLINE_S": "300"})
    took = time.time() - t0
    assert proc.returncode != 0
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout
    failed = json.loads(lines[0])["launcher"]["failed"]
    assert len(failed) == 2 and "deadline of 30 s" in failed[0]["reason"] and "not started" in failed[1]["reason"], failed
    assert took < 75, took


def test_bench_refuses_a_cpu_run_of_the_product_path():
    """No GPU and no test aggregator: an error, not a silent CPU run."""
    env = dict(os.environ)
    env.update({"RGBX_DIST_BACKEND": "gloo", "CUDA_VISIBLE_DEVICES": "", "HIP_VISIBLE_DEVICES": ""})
    env.pop("RGBX_TEST_AGGREGATOR", None)
    env.pop("WORLD_SIZE", None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "T", "--steps", "1"], env=env,
                          capture_output=True, text=True, timeout=300)
    assert proc.returncode != 0 and "no MI355X visible" in proc.stderr


def test_ending_the_launcher_ends_every_rank():
    """SIGTERM (handler) and SIGKILL (parent-death signal) to `python bench.py --gpus N` must not leave supervisors or
    workers behind: on a GPU box a leftover rank would keep its GPU."""
    import signal
    import time
    env = dict(os.environ)
    env.update({"RGBX_DIST_BACKEND": "gloo", "RGBX_TEST_AGGREGATOR": "_dist_worker:OracleAggregator",
                "PYTHONPATH": os.path.join(ROOT, "tests") + os.pathsep + env.get("PYTHONPATH", ""),
                "OMP_NUM_THREADS": "1", "CUDA_VISIBLE_DEVICES": "", "HIP_VISIBLE_DEVICES": "",
                "RGBX_TEST_FAULT": "stall:1:0:first_epoch", "RGBX_LAUNCH_STALL_S": "300"})
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)

    def descendants(pid):
        out = subprocess.run(["ps", "-eo", "pid,ppid"], capture_output=True, text=True).stdout.split("\n")[1:]
        kids = {}
        for line in out:
            if line.strip():
                c, p = (int(v) for v in line.split())
                kids.setdefault(p, []).append(c)
        todo, seen = [pid], []
        while todo:
            for c in kids.get(todo.pop(), []):
                seen.append(c)
                todo.append(c)
        return seen

    for sig in (signal.SIGTERM, signal.SIGKILL):
        marker = f"--steps=2{int(sig)}"  # makes this run's processes recognisable in the process list
        proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "T", marker,
                                 "--warmup", "1"], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        deadline = time.time() + 90
        while time.time() < deadline and len(descendants(proc.pid)) < 5:  # agent, 2 supervisors, 2 workers
            time.sleep(0.5)
        assert len(descendants(proc.pid)) >= 5
        proc.send_signal(sig)
        proc.wait(timeout=60)
        deadline = time.time() + 60
        alive = None
        while time.time() < deadline:
            ps = subprocess.run(["ps", "-eo", "pid,cmd"], capture_output=True, text=True).stdout
            alive = [ln for ln in ps.splitlines() if marker in ln]
            if not alive:
                break
            time.sleep(0.5)
        assert not alive, alive


def test_other_ranks_wait_for_rank_0_while_it_is_alive(tmp_path):
    """supervise._await_rank0: a rank whose own worker is through does not decide the run's end — it waits for rank 0's
    verdict for as long as rank 0's supervisor (`sup.0`) or worker (`attempt{k}.hb.0`) keeps moving, well past the quiet
    limit, and gives up (non-zero exit in supervise) only when rank 0 has been silent for that long or the hard limit
    has passed."""
    import threading
    import time
    from rgb_experiment_amd.dist import supervise as sv
    d = str(tmp_path)
    ok = os.path.join(d, "attempt0.ok")
    # rank 0 silent from the start: None after the quiet limit
    t0 = time.monotonic()
    assert sv._await_rank0(d, 0, [ok], quiet_limit=0.3, hard_limit=5.0) is None
    assert 0.25 < time.monotonic() - t0 < 2.0
    # rank 0 alive (its supervisor keeps touching sup.0) for 4 quiet limits, then the verdict: the waiter is still there
    stop = threading.Event()

    def rank0():
        t_end = time.monotonic() + 1.2
        while time.monotonic() < t_end and not stop.is_set():
            sv._write(os.path.join(d, "sup.0"), "0")
            time.sleep(0.05)
        sv._write(ok, "ok")

    th = threading.Thread(target=rank0)
    th.start()
    try:
        assert sv._await_rank0(d, 0, [ok, os.path.join(d, "gave_up")], quiet_limit=0.3, hard_limit=10.0) == ok
    finally:
        stop.set()
        th.join()
    # alive but never deciding: the hard limit ends the wait
    os.remove(ok)
    sv._write(os.path.join(d, "attempt0.hb.0"), "timed step")
    t0 = time.monotonic()
    assert sv._await_rank0(d, 0, [ok], quiet_limit=30.0, hard_limit=0.4) is None
    assert time.monotonic() - t0 < 2.0
