"""Pin the CPU oracle (oracle/ref_cpu.py) against the golden vectors produced by the reference's own
PyG-free code (tests/golden/make_golden.py) and against hand-derived known answers."""
import math

import numpy as np
import pytest
import torch

from conftest import GOLDEN_GRAPHS
from oracle import ref_cpu as O


def _graph(golden, name):
    ei = torch.from_numpy(golden[f"g1/{name}/edge_index"])
    return int(golden[f"g1/{name}/num_nodes"]), ei


@pytest.mark.parametrize("name", GOLDEN_GRAPHS)
def test_g1_gcn_norm_matches_reference_normalize_adj(golden, name):
    """reference normalize_adj(A+I) holds adj[src,dst] and aggregates at src (itexperiments.py:675,715),
    PyG aggregates at edge_index[1]: the reference matrix equals the oracle's A_hat of the reversed graph."""
    n, ei = _graph(golden, name)
    ref = torch.from_numpy(golden[f"g1/{name}/adj_ref"]).float()
    got = O.gcn_dense_adj(ei.flip(0), n)
    assert torch.allclose(got, ref, atol=1e-6, rtol=0)
    if "undirected" in name:  # symmetric graph: direction does not matter
        assert torch.allclose(O.gcn_dense_adj(ei, n), ref, atol=1e-6, rtol=0)


def test_g1_survey_sample_values(golden):
    """The 4-node sample quoted in SURVEY §8c."""
    ref = golden["g1/survey4/adj_ref"]
    want = np.array([[.5, .4082, 0, 0], [.4082, .3333, .4082, 0], [0, .4082, .5, 0], [.5, 0, 0, .5]])
    assert np.allclose(ref, want, atol=1e-4)


@pytest.mark.parametrize("name", GOLDEN_GRAPHS)
@pytest.mark.parametrize("K", [3, 10])
def test_g2_label_propagation(golden, name, K):
    """itexperiments.py:698-719 restated with the oracle's propagate."""
    n, ei = _graph(golden, name)
    labels = torch.from_numpy(golden[f"g2/{name}/labels"])
    idx = torch.from_numpy(golden[f"g2/{name}/idx"])
    C = int(labels.max()) + 1
    y0 = torch.zeros(n, C)
    y0[idx, labels[idx]] = 1.0
    onehot = torch.nn.functional.one_hot(labels, C).float()
    rei, w = O.gcn_norm(ei.flip(0), None, n)
    y = y0
    for _ in range(K):
        y = O.propagate(rei, y, n, w, "add")
        y[idx] = onehot[idx]
        y = 0.9 * y + 0.1 * y0
    ref = torch.from_numpy(golden[f"g2/{name}/K{K}/out"])
    assert torch.allclose(y, ref, atol=1e-6)


@pytest.mark.parametrize("name", GOLDEN_GRAPHS)
@pytest.mark.parametrize("K,alpha", [(1, 0.1), (10, 0.1), (4, 0.35)])
def test_g3_appnp_matches_pta_inference(golden, name, K, alpha):
    """PTA.inference(h, adj) == APPNP(K, alpha) applied to softmax(h) (models/pta.py:79-84)."""
    n, ei = _graph(golden, name)
    h = torch.from_numpy(golden[f"g3/{name}/h"])
    ref = torch.from_numpy(golden[f"g3/{name}/K{K}_a{alpha}/out"])
    got = O.appnp(torch.softmax(h, dim=-1), ei.flip(0), K, alpha)
    assert torch.allclose(got, ref, atol=1e-6)


@pytest.mark.parametrize("name", GOLDEN_GRAPHS)
def test_next_row_oracles_against_reference_goldens(golden, name):
    """pta_norm_adj_dense == G1, label_propagation == G2, pta_inference == G3 (dense restatements used as
    the oracle of the PTA path)."""
    n, ei = _graph(golden, name)
    adj = O.pta_norm_adj_dense(ei, n)
    assert torch.allclose(adj, torch.from_numpy(golden[f"g1/{name}/adj_ref"]).float(), atol=1e-6)
    labels = torch.from_numpy(golden[f"g2/{name}/labels"])
    idx = torch.from_numpy(golden[f"g2/{name}/idx"])
    for K in (3, 10):
        assert torch.allclose(O.label_propagation(adj, labels, idx, K, 0.1),
                              torch.from_numpy(golden[f"g2/{name}/K{K}/out"]), atol=1e-6)
    h = torch.from_numpy(golden[f"g3/{name}/h"])
    for K, alpha in ((1, 0.1), (10, 0.1), (4, 0.35)):
        assert torch.allclose(O.pta_inference(h, adj, K, alpha),
                              torch.from_numpy(golden[f"g3/{name}/K{K}_a{alpha}/out"]), atol=1e-6)


# ---- hand-derived known answers (independent of the reference) -----------------------------------

def test_kat_gcn_norm_duplicates_and_existing_loops():
    # edges: 0->1 twice, 1->1 (existing loop), 2->0 ; N = 3
    ei = torch.tensor([[0, 0, 1, 2], [1, 1, 1, 0]])
    rei, w = O.gcn_norm(ei, None, 3)
    # rewrite: non-loop edges in order, then one loop per node
    assert rei.tolist() == [[0, 0, 2, 0, 1, 2], [1, 1, 0, 0, 1, 2]]
    deg = [2.0, 3.0, 1.0]  # in-degree incl. the single loop: node0: 2->0 + loop; node1: two 0->1 + loop
    want = [1 / math.sqrt(deg[s] * deg[t]) for s, t in zip(*rei.tolist())]
    assert torch.allclose(w, torch.tensor(want), atol=1e-7)


def test_kat_rewrite_modes_agree_when_unweighted():
    ei = torch.tensor([[0, 1, 1, 2, 2], [1, 1, 2, 2, 0]])
    a, ida = O.rewrite_edges(ei, 4, 1)
    b, idb = O.rewrite_edges(ei, 4, 2)
    assert torch.equal(a, b) and torch.equal(ida, idb)
    assert a.tolist() == [[0, 1, 2, 0, 1, 2, 3], [1, 2, 0, 0, 1, 2, 3]]
    assert ida.tolist() == [0, 2, 4, 5, 6, 7, 8]
    k, idk = O.rewrite_edges(ei, 4, 0)
    assert torch.equal(k, ei) and idk.tolist() == [0, 1, 2, 3, 4]


def test_kat_csr_is_stable():
    ei = torch.tensor([[3, 0, 2, 0, 1], [1, 1, 0, 1, 0]])
    rowptr, col, perm = O.csr_from_edges(ei[1], ei[0], torch.arange(5), 4)
    assert rowptr.tolist() == [0, 2, 5, 5, 5]
    assert col.tolist() == [2, 1, 3, 0, 0]
    assert perm.tolist() == [2, 4, 0, 1, 3]


def test_kat_mean_propagate_isolated_and_duplicates():
    x = torch.tensor([[1., 2.], [3., 4.], [5., 6.]])
    ei = torch.tensor([[0, 0, 1], [1, 1, 1]])  # node 1 <- 0, 0, 1 ; nodes 0 and 2 receive nothing
    out = O.propagate(ei, x, 3, None, "mean")
    assert torch.allclose(out, torch.tensor([[0., 0.], [(1 + 1 + 3) / 3, (2 + 2 + 4) / 3], [0., 0.]]))


def test_kat_my_sage_conv_by_loops():
    torch.manual_seed(0)
    n, f, o = 5, 3, 2
    x = torch.randn(n, f)
    wl, bl, wr, br = torch.randn(o, f), torch.randn(o), torch.randn(o, f), torch.randn(o)
    ei = torch.tensor([[0, 1, 2, 2, 4, 3], [1, 0, 2, 0, 0, 3]])  # two self-loops in the input
    got = O.my_sage_conv(x, ei, wl, bl, wr, br)
    xl = x @ wl.t() + bl
    for i in range(n):
        nbrs = [s for s, t in zip(*ei.tolist()) if t == i and s != i] + [i]
        want = sum(xl[j] for j in nbrs) / len(nbrs) + x[i] @ wr.t() + br
        assert torch.allclose(got[i], want, atol=1e-6)


def test_kat_sage_conv_by_loops():
    torch.manual_seed(1)
    n, f, o = 4, 3, 2
    x = torch.randn(n, f)
    wl, bl, wr = torch.randn(o, f), torch.randn(o), torch.randn(o, f)
    ei = torch.tensor([[0, 1, 2, 2], [1, 0, 2, 0]])  # node 3 isolated, node 2 has only its own loop
    got = O.sage_conv(x, ei, wl, bl, wr)
    for i in range(n):
        nbrs = [s for s, t in zip(*ei.tolist()) if t == i]
        agg = sum(x[j] for j in nbrs) / len(nbrs) if nbrs else torch.zeros(f)
        assert torch.allclose(got[i], agg @ wl.t() + bl + x[i] @ wr.t(), atol=1e-6)


@pytest.mark.parametrize("heads,concat", [(2, True), (3, False), (1, False)])
def test_kat_gat_conv_by_loops(heads, concat):
    torch.manual_seed(2)
    n, f, c = 5, 4, 3
    x = torch.randn(n, f)
    W = torch.randn(heads * c, f)
    a_s, a_d = torch.randn(1, heads, c), torch.randn(1, heads, c)
    bias = torch.randn(heads * c if concat else c)
    ei = torch.tensor([[0, 1, 2, 2, 4, 3, 0], [1, 0, 2, 0, 0, 3, 1]])  # loops + a duplicate edge
    got = O.gat_conv(x, ei, W, a_s, a_d, bias, heads, concat)
    h = (x @ W.t()).view(n, heads, c)
    outs = torch.zeros(n, heads, c)
    for i in range(n):
        nbrs = [s for s, t in zip(*ei.tolist()) if t == i and s != i] + [i]
        for hd in range(heads):
            e = torch.stack([torch.nn.functional.leaky_relu(
                (h[j, hd] * a_s[0, hd]).sum() + (h[i, hd] * a_d[0, hd]).sum(), 0.2) for j in nbrs])
            al = torch.softmax(e, 0)
            outs[i, hd] = sum(a * h[j, hd] for a, j in zip(al, nbrs))
    want = (outs.reshape(n, -1) if concat else outs.mean(1)) + bias
    assert torch.allclose(got, want, atol=1e-5)


def test_kat_appnp_dense():
    torch.manual_seed(3)
    n = 6
    ei = torch.tensor([[0, 1, 2, 3, 4, 5, 0], [1, 2, 3, 4, 5, 0, 3]])
    x = torch.randn(n, 3)
    a = O.gcn_dense_adj(ei, n)
    z = x
    for _ in range(5):
        z = 0.8 * a @ z + 0.2 * x
    assert torch.allclose(O.appnp(x, ei, 5, 0.2), z, atol=1e-6)


def test_batch_norm_matches_torch():
    torch.manual_seed(4)
    bn = torch.nn.BatchNorm1d(7)
    bn.weight.data.uniform_(0.5, 1.5)
    bn.bias.data.uniform_(-1, 1)
    x = torch.randn(50, 7)
    bn.train()
    want = bn(x)
    sd = {"p.weight": bn.weight.data, "p.bias": bn.bias.data, "p.running_mean": bn.running_mean,
          "p.running_var": bn.running_var}
    assert torch.allclose(O.batch_norm(x, sd, "p.", True), want, atol=1e-5)
    bn.eval()
    assert torch.allclose(O.batch_norm(x, sd, "p.", False), bn(x), atol=1e-5)


@pytest.mark.parametrize("aggr", ["add", "mean"])
def test_c_restatement_matches_python_oracle(aggr):
    """oracle/propagate_ref.c (edge-order COO and OpenMP CSR forms) == oracle.propagate."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "oracle")], check=True, capture_output=True)
    g = torch.Generator().manual_seed(5)
    n, e, d = 700, 9000, 37
    ei = torch.randint(0, n, (2, e), generator=g)
    x = torch.randn(n, d, generator=g)
    rei, w = O.gcn_norm(ei, None, n)
    w = w if aggr == "add" else None
    want = O.propagate(rei, x, n, w, aggr)
    assert torch.allclose(O.propagate_c_coo(rei, x, n, w, aggr), want, atol=1e-6)
    rowptr, col, perm = O.csr_from_edges(rei[1], rei[0], torch.arange(rei.size(1)), n)
    ws = None if w is None else w[perm.long()].contiguous()
    for threads in (1, 3):
        assert torch.allclose(O.propagate_c_csr(rowptr, col, ws, x, aggr, threads), want, atol=1e-6)
    # golden G1: the C form reproduces the reference's normalize_adj rows too
    z = np.load(os.path.join(root, "tests", "golden", "reference_pygfree.npz"))
    ei4 = torch.from_numpy(z["g1__survey4__edge_index"]).flip(0)
    r4, w4 = O.gcn_norm(ei4, None, 4)
    dense = O.propagate_c_coo(r4, torch.eye(4), 4, w4, "add")
    assert torch.allclose(dense, torch.from_numpy(z["g1__survey4__adj_ref"]).float(), atol=1e-6)
