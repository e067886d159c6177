"""bench.py as the driver types it: `python bench.py --gpus N` must start its N ranks itself (children, before any
GPU call in the parent) and print ONE JSON line. Rehearsed here on the CPU: gloo collectives, a toy workload, and
the aggregation arithmetic injected from the oracle (the product has no CPU aggregation path)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=600):
    env = dict(os.environ)
    env.update({"RGBX_DIST_BACKEND": "gloo", "RGBX_TEST_AGGREGATOR": "_dist_worker:OracleAggregator",
                "PYTHONPATH": os.path.join(ROOT, "tests") + os.pathsep + env.get("PYTHONPATH", ""),
                "OMP_NUM_THREADS": "1", "CUDA_VISIBLE_DEVICES": "", "HIP_VISIBLE_DEVICES": ""})
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=timeout)
    return proc


@pytest.mark.parametrize("gpus,exchange,scheme", [(2, "reshard", "reshard"), (4, "2x2", "grid2x2"), (2, "halo", "halo"),
                                                  (2, "auto", None), (3, "replicate", "replicate")])
def test_bench_gpus_n_launches_its_own_ranks(gpus, exchange, scheme):
    proc = _run(["--gpus", str(gpus), "--workload", "T", "--steps", "2", "--warmup", "1", "--exchange", exchange])
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


def test_bench_refuses_a_cpu_run_of_the_product_path():
    """No GPU and no test aggregator: an error, not a silent CPU run."""
    env = dict(os.environ)
    env.update({"RGBX_DIST_BACKEND": "gloo", "CUDA_VISIBLE_DEVICES": "", "HIP_VISIBLE_DEVICES": ""})
    env.pop("RGBX_TEST_AGGREGATOR", None)
    env.pop("WORLD_SIZE", None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "T", "--steps", "1"], env=env,
                          capture_output=True, text=True, timeout=300)
    assert proc.returncode != 0 and "no MI355X visible" in proc.stderr
