import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: a long-running case (still part of -m gpu)")


def pytest_collection_modifyitems(config, items):
    """The multi-rank GPU module runs LAST: it is the one part of `-m gpu` that depends on more than this process and the card
    (rank processes, RCCL's socket transport or gloo, rendezvous ports), and under `-x` a failure there must not keep the
    parity modules from running. The BASELINE-size module runs just before it, its S cases before its L cases: the oracle
    halves of those cases are computed on the host cores in the background from session start (oracle_background below),
    while the GPU-bound modules run. (Stable sort: everything else keeps its order.)"""
    def key(it):
        base = it.fspath.basename
        return (base == "test_gpu_dist.py", base == "test_gpu_fullsize.py", base == "test_gpu_fullsize.py" and "L" in
                it.name.partition("[")[2].replace("]", "").split("-"))
    items.sort(key=key)


@pytest.fixture(scope="session", autouse=True)
def oracle_background(request):
    """`-m gpu` sessions on a box with a GPU: two CPU-only processes (the GPU hidden from them) work through the oracle halves
    of the slowest cases (tests/_oracle_jobs.py) while this process runs the GPU-bound modules. Tests fetch a result with
    _oracle_jobs.get(name), which computes the job inline when no background result is (or will be) there — so this
    fixture changes the suite's wall clock and nothing else. RGBX_ORACLE_BG=off disables it."""
    import shutil
    import subprocess
    import tempfile
    wanted = {"test_gpu_fullsize.py", "test_gpu_parity.py"} & {it.fspath.basename for it in request.session.items}
    if not wanted or os.environ.get("RGBX_ORACLE_BG") == "off":
        yield None
        return
    import torch
    if not torch.cuda.is_available():
        yield None
        return
    d = tempfile.mkdtemp(prefix="rgbx_oracle_bg_")
    env = dict(os.environ)
    # CPU only, a share of the host cores each (a one-GPU box has 16): the foreground keeps the rest for its own small oracles
    env.update({"CUDA_VISIBLE_DEVICES": "", "HIP_VISIBLE_DEVICES": "", "OMP_NUM_THREADS": "6", "MKL_NUM_THREADS": "6"})
    groups = ["small", "large"] if "test_gpu_fullsize.py" in wanted else ["small"]
    procs = []
    for group in groups:
        p = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_oracle_jobs.py"), d, group], env=env,
                             stdout=subprocess.DEVNULL, stderr=open(os.path.join(d, f"{group}.err"), "w"))
        with open(os.path.join(d, f"{group}.pid"), "w") as f:
            f.write(str(p.pid))
        procs.append(p)
    os.environ["RGBX_ORACLE_BG"] = d
    yield d
    os.environ.pop("RGBX_ORACLE_BG", None)
    for p in procs:
        if p.poll() is None:
            p.kill()
        p.wait()
    shutil.rmtree(d, ignore_errors=True)


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "reference_pygfree.npz")
    z = np.load(path)
    return {k.replace("__", "/"): z[k] for k in z.files}


GOLDEN_GRAPHS = ["survey4", "path5_undirected", "star6_undirected", "directed7_isolated", "random40_undirected"]
