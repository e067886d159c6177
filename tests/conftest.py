import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: a long-running case (still part of -m gpu)")


def pytest_xdist_auto_num_workers(config):
    """`-n auto` (pytest.ini): workers for a CPU session, NONE for a session that selects GPU tests — those run in this one
    process (at most six processes may hold the card, and the rank processes of the multi-rank module are five of them).
    RGBX_TEST_WORKERS overrides (0 = no workers)."""
    given = os.environ.get("RGBX_TEST_WORKERS")
    if given is not None:
        return int(given)
    expr = (config.getoption("markexpr", "") or "").replace(" ", "")
    cpu_only = "notgpu" in expr and "gpu" not in expr.replace("notgpu", "")
    if not cpu_only:
        return 0
    cpus = usable_cpus()
    return 3 if cpus >= 8 else 2 if cpus >= 4 else 0


def pytest_collection_modifyitems(config, items):
    """Order of a `-m gpu` session: the GPU-bound parity modules, the S cases of the BASELINE-size module, its L cases (their
    oracle halves are computed on the host cores in the background from session start, oracle_background below), the cases
    whose CPU side is heavy, and LAST the multi-rank module — the one part that depends on more than this process and the
    card (rank processes, RCCL's socket transport or gloo, rendezvous ports: under `-x` a failure there must not keep the
    parity modules from running), and the one part that needs the background processes GONE: a process that has run
    autograd's backward has the GPU's device files open (the engine counts the devices of every backend, hidden or not), and
    the box allows six such processes at once — run full12 of round 5 was ended by that guard with four ranks, this process
    and two background processes alive. (Stable sort: everything else keeps its order.)"""
    # L cases in the order the background processes finish their oracle halves (tests/_oracle_jobs.GROUPS)
    l_order = ["test_model_logits_at_sampled_rows", "test_fused_aggregate_transform", "test_appnp_k10",
               "benchmark_size_L[gcn]", "benchmark_size_L[graphsage2]", "benchmark_size_L[appnpstack]",
               "benchmark_size_L[graphsage]", "benchmark_size_L[gat]"]

    def key(it):
        base = it.fspath.basename
        full = base == "test_gpu_fullsize.py"
        at_l = full and ("benchmark_size_L" in it.name or "L" in it.name.partition("[")[2].replace("]", "").split("-"))
        rank = next((i for i, tag in enumerate(l_order) if tag in it.name), len(l_order)) if at_l else 0
        # (this process's thread count stays what the fixture below set: raising it mid-session stalled two runs, full7 / full8)
        late = base == "test_gpu_ingest.py" or any(tag in it.name for tag in _CPU_HEAVY)
        stage = 4 if base == "test_gpu_dist.py" else 3 if late else 2 if at_l else 1 if full else 0
        return (stage, rank)
    items.sort(key=key)


_BG_PROCS = []


def background_oracle_finished(limit_s=240):
    """Block until the background oracle processes have exited (tests/test_gpu_dist.py calls this before its first rank
    process starts); one still running after `limit_s` is killed — its remaining jobs are then computed inline by whoever
    asks for them."""
    import time
    t0 = time.time()
    for p in _BG_PROCS:
        while p.poll() is None:
            if time.time() - t0 > limit_s:
                p.kill()
                p.wait()
                break
            time.sleep(0.2)


def usable_cpus():
    """CPUs this session may keep busy: the cgroup quota where there is one. A one-GPU box of the pool reports 256 from
    os.cpu_count() and grants 16 (`cpu.max` = 1600000 100000): thread counts taken from cpu_count() had 22+ threads throttled
    in three of five scheduler periods (cpu.stat, round 5) — the whole-model fuzz blocks then took 54 s instead of 6."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


_CPU_HEAVY = ("test_spmm_hub_rows_are_split_and_reproducible", "test_csr_build_bit_exact", "test_index_arithmetic_beyond_2_31")
@pytest.fixture(scope="session", autouse=True)
def oracle_background(request):
    """`-m gpu` sessions on a box with a GPU: four CPU-only processes (the GPU hidden from them) work through the oracle halves
    of the slowest cases (tests/_oracle_jobs.py) while this process runs the GPU-bound modules. Tests fetch a result with
    _oracle_jobs.get(name), which computes the job inline when no background result is (or will be) there — so this
    fixture changes the suite's wall clock and nothing else. RGBX_ORACLE_BG=off disables it."""
    import shutil
    import subprocess
    import tempfile
    import torch
    if torch.get_num_threads() > usable_cpus():  # set once, before the first parallel region, and never again (see above)
        torch.set_num_threads(usable_cpus())
    wanted = {"test_gpu_fullsize.py", "test_gpu_parity.py"} & {it.fspath.basename for it in request.session.items}
    if not wanted or os.environ.get("RGBX_ORACLE_BG") == "off" or not torch.cuda.is_available():
        yield None
        return
    d = tempfile.mkdtemp(prefix="rgbx_oracle_bg_")
    env = dict(os.environ)
    # CPU only, a share of the host cores each (a one-GPU box has 16): the foreground keeps the rest for its own small oracles
    env.update({"CUDA_VISIBLE_DEVICES": "", "HIP_VISIBLE_DEVICES": ""})
    groups = ["cora", "small", "large_gcn", "large_mean"] if "test_gpu_fullsize.py" in wanted else ["cora"]
    # the quota split so that everything busy at once stays within it (16: 2 + 3 + 3 + 3 in the background, 5 here; the
    # multi-rank stage's rank processes take 1 thread each and mostly wait for the card)
    cores = usable_cpus()
    share = max(2, (cores - 2) // 5 + 1)
    threads = {g: (2 if g == "cora" else share) for g in groups}
    torch.set_num_threads(max(4, cores - sum(threads.values())))
    procs = []
    for group in groups:
        env_g = dict(env, OMP_NUM_THREADS=str(threads[group]), MKL_NUM_THREADS=str(threads[group]))
        # niced: where this process's own oracle work and the background's meet (the first 200 s), this process goes first —
        # run full9 of round 5 had the whole-model fuzz blocks at 43 / 24 / 18 s instead of 9 s each
        p = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_oracle_jobs.py"), d, group], env=env_g,
                             stdout=subprocess.DEVNULL, stderr=open(os.path.join(d, f"{group}.err"), "w"),
                             preexec_fn=lambda: os.nice(10))
        with open(os.path.join(d, f"{group}.pid"), "w") as f:
            f.write(str(p.pid))
        procs.append(p)
        _BG_PROCS.append(p)
    os.environ["RGBX_ORACLE_BG"] = d
    yield d
    os.environ.pop("RGBX_ORACLE_BG", None)
    del _BG_PROCS[:]
    for p in procs:
        if p.poll() is None:
            p.kill()
        p.wait()
    out = os.path.join(ROOT, "gpurun_out")  # the jobs' timings (seconds per job) next to the run's other scratch records
    if os.path.isdir(out):
        for group in groups:
            try:
                shutil.copy(os.path.join(d, f"{group}.log"), os.path.join(out, f"oracle_bg_{group}.log"))
            except OSError:
                pass
    shutil.rmtree(d, ignore_errors=True)


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "reference_pygfree.npz")
    z = np.load(path)
    return {k.replace("__", "/"): z[k] for k in z.files}


GOLDEN_GRAPHS = ["survey4", "path5_undirected", "star6_undirected", "directed7_isolated", "random40_undirected"]
