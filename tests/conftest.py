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
    parity modules from running. (Stable sort: everything else keeps its order.)"""
    items.sort(key=lambda it: it.fspath.basename == "test_gpu_dist.py")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "reference_pygfree.npz")
    z = np.load(path)
    return {k.replace("__", "/"): z[k] for k in z.files}


GOLDEN_GRAPHS = ["survey4", "path5_undirected", "star6_undirected", "directed7_isolated", "random40_undirected"]
