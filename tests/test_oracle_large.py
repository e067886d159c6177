"""oracle/large.py (the checker of the BASELINE-size gradient tests: propagate through the C restatement over a CSR, its
adjoint over the transposed CSR) against oracle/ref_cpu.py (the PyG dataflow under torch autograd, pinned by the golden
vectors), on graphs ref_cpu can hold: train-mode logits, loss and EVERY parameter gradient of the four models."""
import pytest
import torch

from oracle import large as OL
from oracle import ref_cpu as O

CASES = {
    "gcn": (lambda sd, x, ei, tr: O.gcn_forward(sd, x, ei, 2, tr), dict(num_layers=2)),
    "graphsage": (lambda sd, x, ei, tr: O.graphsage_forward(sd, x, ei, 2, tr), dict(num_layers=2)),
    "graphsage2": (lambda sd, x, ei, tr: O.graphsage2_forward(sd, x, ei, 2, tr), dict(num_layers=2)),
    "appnpstack": (lambda sd, x, ei, tr: O.appnp_stack_forward(sd, x, ei, 10, 0.1, tr), dict(K=10, alpha=0.1)),
    # (GATConv: ref_cpu's own statement is unpinnable here — PyG semantics, nothing in the reference to check it against)
    "gat": (lambda sd, x, ei, tr: O.gat_forward(sd, x, ei, 2, 4, tr), dict(num_layers=2, heads=4)),
}


def _state(name, f, hid, c, gen):
    r = lambda *s: torch.randn(*s, generator=gen) / max(s[-1], 1) ** 0.5
    bn = lambda p, d: {p + "weight": 1 + 0.1 * r(d), p + "bias": 0.1 * r(d), p + "running_mean": torch.zeros(d),
                       p + "running_var": torch.ones(d), p + "num_batches_tracked": torch.zeros((), dtype=torch.long)}
    if name == "gcn":
        sd = {"convs.0.lin.weight": r(hid, f), "convs.0.bias": 0.1 * r(hid), "convs.1.lin.weight": r(c, hid),
              "convs.1.bias": 0.1 * r(c)}
        sd.update(bn("bns.0.", hid))
    elif name == "graphsage":
        sd = {}
        for i, (o, k) in enumerate([(hid, f), (c, hid)]):
            sd.update({f"convs.{i}.lin_l.weight": r(o, k), f"convs.{i}.lin_l.bias": 0.1 * r(o),
                       f"convs.{i}.lin_r.weight": r(o, k), f"convs.{i}.lin_r.bias": 0.1 * r(o)})
        sd.update(bn("bns.0.", hid))
    elif name == "graphsage2":
        sd = {}
        for i, (o, k) in enumerate([(hid, f), (c, hid)]):
            sd.update({f"convs.{i}.lin_l.weight": r(o, k), f"convs.{i}.lin_l.bias": 0.1 * r(o),
                       f"convs.{i}.lin_r.weight": r(o, k)})
        sd.update(bn("bns.0.", hid))
    elif name == "gat":  # 4 heads x hid / 4 channels, then one head of c
        H = 4
        sd = {"convs.0.lin_src.weight": r(hid, f), "convs.0.att_src": r(1, H, hid // H), "convs.0.att_dst": r(1, H, hid // H),
              "convs.0.bias": 0.1 * r(hid), "convs.1.lin_src.weight": r(c, hid), "convs.1.att_src": r(1, 1, c),
              "convs.1.att_dst": r(1, 1, c), "convs.1.bias": 0.1 * r(c)}
        sd.update(bn("bns.0.", hid))
    else:
        sd = {"lin1.weight": r(hid, f), "lin1.bias": 0.1 * r(hid), "lin2.weight": r(c, hid), "lin2.bias": 0.1 * r(c)}
        sd.update(bn("bn.", hid))
    return sd


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("n,e", [(300, 2500), (64, 40), (1, 0)])
def test_large_graph_checker_equals_the_pinned_oracle(name, n, e):
    gen = torch.Generator().manual_seed(n + e)
    ei = torch.randint(0, n, (2, e), generator=gen)
    if e:  # self-loops and duplicates, the rewrite rules' cases
        ei = torch.cat([ei, ei[:, :5], torch.arange(min(n, 4)).repeat(2, 1)], dim=1)
    f, hid, c = 12, 16, 5
    x = torch.randn(n, f, generator=gen)
    y = torch.randint(0, c, (n,), generator=gen)
    mask = torch.rand(n, generator=gen) < 0.6
    mask[0] = True
    sd = _state(name, f, hid, c, gen)
    fwd, kw = CASES[name]
    graph = OL.graphs_for(name, ei, n, threads=2)
    loss, grads, emb = OL.loss_and_grads(name, sd, x, y, mask, graph, **kw)
    ref_sd = {k: v.clone().requires_grad_(v.is_floating_point() and "running_" not in k) for k, v in sd.items()}
    ref = fwd(ref_sd, x, ei, True)
    ref_loss = OL.masked_nll(ref, y, mask)
    ref_loss.backward()
    assert (emb - ref["emb"].detach()).abs().max().item() < 2e-5
    assert abs(loss - ref_loss.item()) < 1e-5
    assert set(grads) == {k for k, v in ref_sd.items() if v.requires_grad}
    for k, g in grads.items():
        rg = ref_sd[k].grad
        assert (g - rg).abs().max().item() < 2e-5 * max(1.0, rg.abs().max().item()), k
    # eval mode too (running statistics)
    ev = OL.forward(name, sd, x, graph, False, **kw)["emb"]
    assert (ev - fwd(sd, x, ei, False)["emb"]).abs().max().item() < 2e-5


def test_shared_and_c_grouped_structures_equal_the_plain_ones():
    """Two speed-ups of the BASELINE-size checker leave its structures unchanged: the 'gat' graph derived from a 'mean' graph
    with loops_mode 2 (CsrGraph.as_gat: shared CSRs) equals the one built directly, and csr_from_edges' C counting sort
    (taken from 2^20 keys on) returns what torch.sort(stable=True) returns."""
    gen = torch.Generator().manual_seed(3)
    n, e = 5000, 60000
    ei = torch.randint(0, n, (2, e), generator=gen)
    ei = torch.cat([ei, ei[:, :50], torch.arange(40).repeat(2, 1)], dim=1)
    direct = OL.CsrGraph(ei, n, "gat", threads=2)
    shared = OL.CsrGraph(ei, n, "mean", loops_mode=2, threads=2).as_gat()
    for name in ("rowptr", "col", "rowptr_t", "col_t", "fwd_slot"):
        assert torch.equal(getattr(direct, name), getattr(shared, name)), name
    assert shared.kind == "gat" and shared.w is None and shared.nnz == direct.nnz
    m = (1 << 20) + 17
    key = torch.randint(0, 70000, (m,), generator=gen)
    other, ids = torch.randint(0, 70000, (m,), generator=gen), torch.arange(m)
    rowptr, col, perm = O.csr_from_edges(key, other, ids, 70000)  # C path
    order = torch.sort(key, stable=True)[1]
    ptr = torch.zeros(70001, dtype=torch.int64)
    ptr[1:] = torch.cumsum(torch.bincount(key, minlength=70000), 0)
    assert torch.equal(rowptr, ptr.int()) and torch.equal(col, other[order].int()) and torch.equal(perm, ids[order].int())
