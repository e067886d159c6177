import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: a long-running case (still part of -m gpu)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "reference_pygfree.npz")
    z = np.load(path)
    return {k.replace("__", "/"): z[k] for k in z.files}


GOLDEN_GRAPHS = ["survey4", "path5_undirected", "star6_undirected", "directed7_isolated", "random40_undirected"]
