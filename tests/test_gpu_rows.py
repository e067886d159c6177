"""The layers as the reference's DEFAULT shapes run them (in > out: F -> hidden 64 -> C in {3, 6, 7, 40, 47},
reference initial_params.py:25-29, 最终结果.csv): transform first, gather at the output width, BatchNorm's column sums
and the masked cross-entropy taken inside the gather kernel (rgbx_spmm_csr_epilogue_f32). HIP path vs the CPU oracle.

Tolerances: logits / losses 1e-4 absolute (the north-star bound) or tighter where stated; statistics counts exact; the
epilogue kernel's output rows are BIT-identical to rgbx_spmm_csr_f32's (same summation order)."""
import pytest
import torch

from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def graph_with_isolated_nodes(n, e, seed, hub=0):
    """Random directed edges among the first 90 % of the nodes' targets: the last tenth has no in-edge (isolated as
    aggregation targets), node n - 1 has no edge at all; a few self-loops and duplicates; optionally one hub target."""
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(0, n - 1, (e,), generator=g)
    dst = torch.randint(0, max(1, n - n // 10), (e,), generator=g)
    ei = torch.stack([src, dst])
    k = torch.randint(0, n - 1, (7,), generator=g)
    ei = torch.cat([ei, torch.stack([k, k]), ei[:, :5]], dim=1)
    if hub:
        ei = torch.cat([ei, torch.stack([torch.randint(0, n - 1, (hub,), generator=g), torch.full((hub,), 3)])], dim=1)
    return ei


def _kind_graph(dev, ei, n, kind):
    from rgb_experiment_amd.graph import Graph
    return Graph(ei.to(dev), n, {"gcn": 1, "mean": 0, "sum": 0}[kind])


@pytest.mark.parametrize("d", [4, 8, 40, 64, 128, 256])
@pytest.mark.parametrize("kind", ["gcn", "mean", "sum"])
def test_row_kernel_column_sums(dev, d, kind, monkeypatch):
    """out is bit-identical to the plain row gather's; colsums = [sum out, sum out^2] per column (fp64 reference of the
    same out); with the additive operand, a bias, isolated targets and a hub row cut into chunks."""
    from rgb_experiment_amd import graph as G
    from rgb_experiment_amd import ops
    monkeypatch.setattr(G, "LONG_ROW_SLOTS", 256)
    n = 4133  # not a multiple of the 32-row tile
    ei = graph_with_isolated_nodes(n, 30000, d, hub=1500)
    g = _kind_graph(dev, ei, n, kind)
    assert g.fwd.split is not None
    gen = torch.Generator().manual_seed(d + 1)
    h = torch.randn(n, 2 * d, generator=gen).to(dev)
    bias = torch.randn(d, generator=gen).to(dev)
    w, rs = ops._kind_weights(g, kind)
    for y, b in ((None, None), (h[:, d:], bias)):
        want = ops.spmm_raw(g.fwd, w, rs, h[:, :d], y=y, a=1.0, b=1.0, bias=b)
        out, cs = ops.spmm_epilogue_raw(g.fwd, w, rs, h[:, :d], y=y, a=1.0, b=1.0, bias=b, want_colsums=True)
        assert torch.equal(out, want)
        ref = torch.stack([want.double().sum(0), (want.double() ** 2).sum(0)])
        assert (cs - ref).abs().max().item() < 1e-6 * max(1.0, ref.abs().max().item())
        again = ops.spmm_epilogue_raw(g.fwd, w, rs, h[:, :d], y=y, a=1.0, b=1.0, bias=b, want_colsums=True)[1]
        assert torch.equal(cs, again)  # fixed summation order


@pytest.mark.parametrize("n", [2, 3, 33, 4133])
@pytest.mark.parametrize("route", ["rows", "fused", "dense"])
def test_column_statistics_of_nearly_constant_columns(dev, n, route):
    """The variance a BatchNorm takes from handed-over column sums, on columns whose spread is far below their mean (means
    0.5 .. 3, standard deviations 1e-4 .. 1): within 1e-4 of var + eps of the float64 statistics of the same output, from
    the row kernel's epilogue, from the fused kernel's MFMA tiles and from the dense launch. A per-tile record of
    (sum x, sum x^2) in float32 lost it: x^2 rounds at 6e-8 mean^2, the difference var = E[x^2] - mean^2 of a column with
    std 4e-4 was off by a third, and whole-model fuzz seed 557 (graphsage2, 2 nodes, hidden 100: the first shape whose
    statistics came from the row kernel) had its logits 2e-2 off; the records now hold the squared deviations from the
    tile's own mean."""
    from rgb_experiment_amd import nn as RN
    from rgb_experiment_amd import ops
    d = 128
    gen = torch.Generator().manual_seed(n)
    mean = torch.rand(d, generator=gen) * 2.5 + 0.5
    std = 10 ** (-4 * torch.rand(d, generator=gen))
    x = (mean + std * torch.randn(n, d, generator=gen)).to(dev)
    ei = torch.arange(n).repeat(2, 1)  # every node aggregates itself: the layer's output is x
    g = _kind_graph(dev, ei, n, "sum")
    if route == "rows":
        out, cs = ops.spmm_epilogue_raw(g.fwd, None, None, x, want_colsums=True)
    else:
        eye = torch.eye(d, device=dev)
        out, cs = _fused_with_sums(ops, x, eye, g.fwd if route == "fused" else None)
    assert torch.equal(out, x)
    xd = x.double()
    var64 = xd.var(0, unbiased=False)
    got_mean = cs[0] / n
    got_var = cs[1] / n - got_mean ** 2
    assert (got_mean - xd.mean(0)).abs().max().item() < 1e-6
    rel = ((got_var - var64).abs() / (var64 + 1e-5)).max().item()
    assert rel < 1e-4, (rel, n, route)
    bn_a, bn_b = RN.BatchNorm1d(d).to(dev), RN.BatchNorm1d(d).to(dev)
    ya, yb = bn_a(x, colsums=cs), bn_b(x)
    assert (ya - yb).abs().max().item() < 2e-4 * max(1.0, yb.abs().max().item())


def _fused_with_sums(ops, x, wt, csr):
    res = ops.fused_layer(x, wt, csr=csr, want_colsums=True)
    return res[0], res[-1]


@pytest.mark.parametrize("C", [1, 3, 7, 40, 47, 128, 130, 256])
@pytest.mark.parametrize("kind", ["gcn", "mean"])
def test_row_kernel_cross_entropy(dev, C, kind, monkeypatch):
    """Statistics and loss gradient of the masked cross-entropy from inside the gather, against the loss kernels run on
    the materialised logits (rgbx_masked_ce_fwd_f32 / _bwd_f32, themselves checked against torch in test_gpu_parity):
    padded class counts, labels out of range, isolated targets, a hub row, two masks from one launch."""
    from rgb_experiment_amd import graph as G
    from rgb_experiment_amd import ops
    monkeypatch.setattr(G, "LONG_ROW_SLOTS", 256)
    n = 3001
    d = (C + 3) // 4 * 4
    ei = graph_with_isolated_nodes(n, 25000, C, hub=1200)
    g = _kind_graph(dev, ei, n, kind)
    gen = torch.Generator().manual_seed(C)
    h = torch.randn(n, 2 * d, generator=gen)
    h[:, C:d] = 0  # pad columns, as pad_rows4 produces them
    h = h.to(dev)
    bias = torch.nn.functional.pad(torch.randn(C, generator=gen), (0, d - C)).to(dev)
    y = torch.randint(0, C, (n,), generator=gen)
    y[5], y[6] = -1, C  # unlabelled / out of range: deselected
    mask = torch.rand(n, generator=gen) < 0.5
    mask[3] = True  # the hub row is selected
    mask[n - 1] = True  # a node without any edge is selected
    mask2 = torch.rand(n, generator=gen) < 0.3
    y, mask, mask2 = y.to(dev), mask.to(dev), mask2.to(dev)
    w, rs = ops._kind_weights(g, kind)
    logits = ops.spmm_raw(g.fwd, w, rs, h[:, :d], y=h[:, d:], a=1.0, b=1.0, bias=bias)[:, :C].contiguous()
    want = ops.masked_ce_accuracy(logits, y, mask)
    none, stats = ops.spmm_epilogue_raw(g.fwd, w, rs, h[:, :d], y=h[:, d:], a=1.0, b=1.0, bias=bias, ce=(y, mask, None),
                                        n_classes=C)
    assert none is None
    assert torch.equal(stats[1:], want[1:])
    assert abs(stats[0].item() - want[0].item()) < 1e-5 * max(1.0, want[0].item())
    # no mask = every labelled row
    _, stats_all = ops.spmm_epilogue_raw(g.fwd, w, rs, h[:, :d], y=h[:, d:], a=1.0, b=1.0, bias=bias, ce=(y, None, None),
                                         n_classes=C)
    want_all = ops.masked_ce_accuracy(logits, y, None)
    assert torch.equal(stats_all[1:], want_all[1:]) and int(stats_all[1].item()) == n - 2
    assert abs(stats_all[0].item() - want_all[0].item()) < 1e-5 * max(1.0, want_all[0].item())
    # two masks, one launch
    _, pair = ops.spmm_epilogue_raw(g.fwd, w, rs, h[:, :d], y=h[:, d:], a=1.0, b=1.0, bias=bias, ce=(y, (mask, mask2), None),
                                    n_classes=C)
    want2 = ops.masked_ce_accuracy(logits, y, mask2)
    assert pair.shape == (2, 3) and torch.equal(pair[0], stats)
    assert torch.equal(pair[1, 1:], want2[1:]) and abs(pair[1, 0].item() - want2[0].item()) < 1e-5 * max(1.0, want2[0].item())
    # loss gradient in place of the logits
    scale = ops.mask_scale(y, mask, C)
    grad, stats_g = ops.spmm_epilogue_raw(g.fwd, w, rs, h[:, :d], y=h[:, d:], a=1.0, b=1.0, bias=bias, ce=(y, mask, scale),
                                          n_classes=C)
    assert torch.equal(stats_g, stats)
    lg = logits.clone().requires_grad_(True)
    ops.masked_ce_loss(lg, y, mask).backward()
    assert (grad[:, :C] - lg.grad).abs().max().item() < 1e-6
    assert torch.equal(grad[:, C:], torch.zeros(n, d - C, device=dev))
    sel = (mask & (y >= 0) & (y < C))
    assert torch.equal(grad[~sel], torch.zeros_like(grad[~sel]))


def test_row_kernel_argument_checks(dev):
    from rgb_experiment_amd import ops
    n = 100
    g = _kind_graph(dev, graph_with_isolated_nodes(n, 500, 0), n, "gcn")
    y = torch.zeros(n, dtype=torch.int64, device=dev)
    with pytest.raises(RuntimeError, match="d %% 4|pad the rows|d % 4"):
        ops.spmm_epilogue_raw(g.fwd, g.w, None, torch.zeros(n, 7, device=dev), want_colsums=True)
    with pytest.raises(RuntimeError, match="256"):
        ops.spmm_epilogue_raw(g.fwd, g.w, None, torch.zeros(n, 260, device=dev), want_colsums=True)
    with pytest.raises(RuntimeError, match="n_classes"):
        ops.spmm_epilogue_raw(g.fwd, g.w, None, torch.zeros(n, 8, device=dev), ce=(y, None, None), n_classes=9)
    with pytest.raises(RuntimeError, match="no epilogue"):
        ops.spmm_epilogue_raw(g.fwd, g.w, None, torch.zeros(n, 8, device=dev))
    with pytest.raises(RuntimeError, match="int64"):
        ops.spmm_epilogue_raw(g.fwd, g.w, None, torch.zeros(n, 8, device=dev), ce=(y.int(), None, None), n_classes=7)


# ---- whole models at the reference's default shapes ------------------------------------------------------------------

def _default_shape_case(name):
    from rgb_experiment_amd import models as M
    kw = dict(num_layers=2, hidden_unit=64, dropout_rate=0.5)  # initial_params.py:25-29
    if name == "gcn":
        return M.GCN, kw, lambda sd, x, ei, tr: O.gcn_forward(sd, x, ei, 2, tr), "_PropagateRowsBackward"
    if name == "graphsage":
        return M.GraphSAGE, kw, lambda sd, x, ei, tr: O.graphsage_forward(sd, x, ei, 2, tr), "_PropagateRowsBackward"
    if name == "graphsage2":
        return M.GraphSAGE2, kw, lambda sd, x, ei, tr: O.graphsage2_forward(sd, x, ei, 2, tr), "_PropagateRowsBackward"
    if name == "appnpstack":
        return (M.APPNPStack, dict(hidden_unit=64, K=10, alpha=0.1, dropout_rate=0.5),
                lambda sd, x, ei, tr: O.appnp_stack_forward(sd, x, ei, 10, 0.1, tr), "_APPNPCEBackward")
    raise KeyError(name)


class _GatherSpy:
    """Records (width, epilogue?, wrote an output?) of every aggregation launch."""

    def __init__(self, monkeypatch):
        from rgb_experiment_amd import ops
        self.calls = []
        raw, epi, appnp = ops.spmm_raw, ops.spmm_epilogue_raw, ops.appnp_raw

        def spy_raw(csr, w, rs, x, *a, **k):
            if str(k.get("kind", "")).startswith("features"):  # x W^T over the features' non-zeros: not a graph gather
                self.feature_products = getattr(self, "feature_products", 0) + 1
            else:
                self.calls.append((x.size(1), None, True))
            return raw(csr, w, rs, x, *a, **k)

        def spy_epi(csr, w, rs, x, *a, **k):
            res = epi(csr, w, rs, x, *a, **k)
            self.calls.append((x.size(1), "ce" if k.get("ce") is not None else "colsums", res[0] is not None))
            return res

        def spy_appnp(csr, w, h, *a, **k):
            self.calls.append((h.size(1), None, True))
            return appnp(csr, w, h, *a, **k)

        monkeypatch.setattr(ops, "spmm_raw", spy_raw)
        monkeypatch.setattr(ops, "appnp_raw", spy_appnp)
        monkeypatch.setattr(ops, "spmm_epilogue_raw", spy_epi)


@pytest.mark.parametrize("name", ["gcn", "graphsage", "graphsage2", "appnpstack"])
@pytest.mark.parametrize("C", [3, 7, 40, 47])
def test_models_at_the_reference_default_shapes(dev, name, C, monkeypatch):
    """F = 1433 -> hidden 64 -> C: loss, statistics and every parameter gradient of one training step against the
    oracle under autograd; eval statistics of both masks from one forward against the oracle's logits; every gather runs
    at width <= 64 (graphsage2: not at F = 1433, the row the reference marks OOM, README.md:74) and the eval forward
    writes no [N, C] matrix."""
    cls, kw, oracle_fwd, grad_fn = _default_shape_case(name)
    n, f = 2708, 1433
    ei = graph_with_isolated_nodes(n, 10556, C)
    gen = torch.Generator().manual_seed(C)
    x = (torch.rand(n, f, generator=gen) < 0.0126).float()  # bag-of-words-like: ~18 ones per row
    x = x / x.sum(1, keepdim=True).clamp(min=1)
    y = torch.randint(0, C, (n,), generator=gen)
    mask = torch.rand(n, generator=gen) < 0.6
    mask[n - 1] = True  # a node without any edge
    mask_b = ~mask
    torch.manual_seed(14530529)
    model = cls(input_dim=f, output_dim=C, **kw)
    with torch.no_grad():
        for k, p in model.named_parameters():
            if p.dim() == 1:
                p.uniform_(-0.5, 0.5) if ("bn" not in k or "bias" in k) else p.uniform_(0.5, 1.5)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    params = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd0.items()}
    out_o = oracle_fwd(params, x, ei, True)["out"]
    loss_o = torch.nn.functional.nll_loss(out_o[mask], y[mask])
    loss_o.backward()

    spy = _GatherSpy(monkeypatch)
    model.to(dev).train()
    xd, eid, yd, md, mbd = x.to(dev), ei.to(dev), y.to(dev), mask.to(dev), mask_b.to(dev)
    from rgb_experiment_amd import ops
    if C == 47:  # rows on 16-byte boundaries (stride 1436), dense products
        xd = ops.align_rows(xd)
        assert xd.stride(0) == 1436 and xd.shape == (n, f)
    elif C in (3, 7):  # what experiment() does with such features: aligned rows + products over the non-zeros alone
        xd = ops.prepare_features(xd)
        assert xd.stride(0) == 1436 and getattr(xd, "_rgbx_sparse", None) is not None
    loss, stats = model.masked_ce(xd, eid, yd, md)
    assert type(loss.grad_fn).__name__ == grad_fn  # the loss came out of the gather kernel
    loss.backward()
    assert abs(loss.item() - loss_o.item()) < 1e-5
    assert int(stats[1].item()) == int(mask.sum())
    assert int(stats[2].item()) == int((out_o[mask].argmax(1) == y[mask]).sum())
    for k, p in model.named_parameters():
        want = params[k].grad
        assert (p.grad.cpu() - want).abs().max().item() < 1e-4 * max(1.0, want.abs().max().item()), k
    assert max(c[0] for c in spy.calls) <= 64, spy.calls
    assert spy.calls and any(c[1] == "ce" for c in spy.calls)

    # eval: both masks from one forward, no logits written; BatchNorm's running statistics moved once, as the oracle's did
    sd1 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    ref_sd = dict(sd0)
    for k in sd1:
        if "running_" in k or "num_batches" in k:
            ref_sd[k] = sd1[k]
    ref = oracle_fwd(ref_sd, x, ei, False)
    spy.calls.clear()
    model.eval()
    with torch.no_grad():
        model.load_state_dict({k: v.to(dev) for k, v in ref_sd.items()})
        pair = model.masked_ce_pair(xd, eid, yd, md, mbd)
        emb = model(xd, eid)["emb"]
    last = [c for c in spy.calls if c[1] == "ce"]
    assert len(last) == 1 and last[0][2] is False  # one loss launch, which wrote nothing
    assert (emb.cpu() - ref["emb"]).abs().max().item() < TOL
    for row, m in ((pair[0], mask), (pair[1], mask_b)):
        nll = torch.nn.functional.nll_loss(ref["out"][m], y[m], reduction="sum").item()
        assert int(row[1].item()) == int(m.sum())
        assert abs(row[0].item() - nll) < 1e-4 * max(1.0, nll)
        # arg-max ties aside (none on random data), the hit counts agree
        assert abs(int(row[2].item()) - int((ref["emb"][m].argmax(1) == y[m]).sum())) <= 1


@pytest.mark.parametrize("name", ["gcn", "graphsage", "graphsage2"])
def test_hidden_layer_hands_batchnorm_its_column_sums_from_the_row_kernel(dev, name, monkeypatch):
    """in > out hidden layer followed by a training-mode BatchNorm: the gather kernel's column sums replace the statistics
    pass; outputs, running statistics and gradients equal the route with the statistics pass."""
    from rgb_experiment_amd import ops
    cls, kw, _, _ = _default_shape_case(name)
    kw = dict(kw, num_layers=3)
    n, f, C = 3000, 200, 16
    ei = graph_with_isolated_nodes(n, 20000, 9)
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(n, f, generator=gen).to(dev)
    y = torch.randint(0, C, (n,), generator=gen).to(dev)
    torch.manual_seed(5)
    model = cls(input_dim=f, output_dim=C, **kw).to(dev).train()
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    spy = _GatherSpy(monkeypatch)
    out = model(x, ei.to(dev))["emb"]
    torch.nn.functional.cross_entropy(out, y).backward()
    assert any(c[1] == "colsums" for c in spy.calls), spy.calls
    got = {k: p.grad.clone() for k, p in model.named_parameters()}
    sd1 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    monkeypatch.setattr(ops, "rows_epilogue_ok", lambda *a, **k: False)
    model.load_state_dict(sd0)
    model.zero_grad()
    spy.calls.clear()
    out2 = model(x, ei.to(dev))["emb"]
    torch.nn.functional.cross_entropy(out2, y).backward()
    assert not any(c[1] for c in spy.calls)
    assert (out - out2).abs().max().item() < 1e-5
    for k, p in model.named_parameters():
        assert (p.grad - got[k]).abs().max().item() < 1e-5 * max(1.0, got[k].abs().max().item()), k
    for k, v in model.state_dict().items():
        if "running_" in k:
            assert (v - sd1[k]).abs().max().item() < 1e-6, k


@pytest.mark.parametrize("K", [1, 2, 10])
@pytest.mark.parametrize("C", [7, 40])
def test_appnp_with_the_loss_in_its_last_step(dev, K, C):
    """ops.appnp_propagate_ce against APPNP followed by the loss kernels on its logits: loss, statistics, input gradient."""
    from rgb_experiment_amd import ops
    from rgb_experiment_amd.graph import Graph
    n, d = 5000, (C + 3) // 4 * 4
    ei = graph_with_isolated_nodes(n, 40000, K + C)
    g = Graph(ei.to(dev), n, 1)
    gen = torch.Generator().manual_seed(K)
    h0 = torch.nn.functional.pad(torch.randn(n, C, generator=gen), (0, d - C)).to(dev)
    y = torch.randint(0, C, (n,), generator=gen).to(dev)
    mask = (torch.rand(n, generator=gen) < 0.5).to(dev)
    h = h0.clone().requires_grad_(True)
    loss, stats = ops.appnp_propagate_ce(h, g, K, 0.1, C, y, mask)
    loss.backward()
    h2 = h0.clone().requires_grad_(True)
    logits = ops.appnp_propagate(h2, g, K, 0.1)[:, :C]
    want, wstats = ops.masked_ce_loss(logits, y, mask, with_stats=True)
    want.backward()
    assert abs(loss.item() - want.item()) < 1e-6 and torch.equal(stats[1:], wstats[1:])
    assert (h.grad - h2.grad).abs().max().item() < 1e-6
    ref = O.appnp(h0[:, :C].cpu().double(), ei, K, 0.1)
    sel = mask.cpu()
    nll = torch.nn.functional.cross_entropy(ref[sel], y.cpu()[sel]).item()
    assert abs(loss.item() - nll) < 1e-5


def test_gin_and_uncached_sgc_transform_first(dev, monkeypatch):
    """GINConv's first Linear with in > out and SGConv(cached=False) with in > out gather at the output width; logits and
    gradients against the oracle."""
    from rgb_experiment_amd import models as M
    n, f, C = 2000, 300, 7
    ei = graph_with_isolated_nodes(n, 12000, 4)
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(n, f, generator=gen)
    y = torch.randint(0, C, (n,), generator=gen)
    for name in ("gin", "sgc"):
        torch.manual_seed(2)
        if name == "gin":
            model = M.GIN(input_dim=f, output_dim=C, hidden_unit=64, num_layers=2, dropout_rate=0.0)
            fwd = lambda sd, tr: O.gin_forward(sd, x, ei, 2, tr)
        else:
            model = M.SGC(input_dim=f, output_dim=C, K=2, cached=False)
            fwd = lambda sd, tr: O.sgc_forward(sd, x, ei, 2)
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        params = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
        ref = fwd(params, True)
        torch.nn.functional.nll_loss(ref["out"], y).backward()
        spy = _GatherSpy(monkeypatch)
        model.to(dev).train()
        out = model(x.to(dev), ei.to(dev))
        torch.nn.functional.nll_loss(out["out"], y.to(dev)).backward()
        assert max(c[0] for c in spy.calls) <= 64, (name, spy.calls)
        assert (out["emb"].detach().cpu() - ref["emb"].detach()).abs().max().item() < TOL, name
        for k, p in model.named_parameters():
            want = params[k].grad
            assert (p.grad.cpu() - want).abs().max().item() < 1e-4 * max(1.0, want.abs().max().item()), (name, k)


@pytest.mark.parametrize("K,M,N", [(2708, 64, 1433), (5000, 7, 1433), (4099, 128, 1433), (3000, 47, 66), (70000, 64, 130),
                                   (1, 3, 5)])
def test_gemm_tn_reads_aligned_rows_with_odd_widths(dev, K, M, N):
    """dW = dY^T X for operands kept as [:, :n] views of 16-byte-row buffers (ops.align_rows' layout): the kernel's float4
    path reads into the rows' padding, whose contents (NaN here) must not reach any stored product."""
    from rgb_experiment_amd import ops
    gen = torch.Generator().manual_seed(K + M + N)
    a, b = torch.randn(K, M, generator=gen), torch.randn(K, N, generator=gen)
    want = a.double().t() @ b.double()

    def padded(t):
        base = torch.full((t.size(0), (t.size(1) + 3) // 4 * 4), float("nan"), device=dev)
        base[:, :t.size(1)] = t.to(dev)
        return base[:, :t.size(1)]

    ad, bd = padded(a), padded(b)
    got, sums = ops.gemm_tn(ad, bd, colsum=True)
    scale = max(1.0, want.abs().max().item())
    assert torch.isfinite(got).all()
    assert (got.cpu().double() - want).abs().max().item() < 2e-5 * scale * max(1.0, (K / 1000) ** 0.5)
    assert (sums.cpu().double() - a.double().sum(0)).abs().max().item() < 1e-3
    again = ops.gemm_tn(ad, bd)
    assert torch.equal(again, got)  # slab-ordered reduction: reproducible
    # a view whose storage ends before the last row's padding is copied, not over-read
    if N % 4 and K > 1:
        tight = torch.empty(K * ((N + 3) // 4 * 4) - 1, device=dev)[: (K - 1) * ((N + 3) // 4 * 4) + N]
        v = tight.as_strided((K, N), ((N + 3) // 4 * 4, 1))
        v.copy_(b.to(dev))
        assert not ops._rows_padded_readable(v)
        assert (ops.gemm_tn(ad, v).cpu().double() - want).abs().max().item() < 2e-5 * scale * max(1.0, (K / 1000) ** 0.5)


def test_align_rows_layout(dev):
    from rgb_experiment_amd import ops
    x = torch.randn(1000, 1433, device=dev)
    v = ops.align_rows(x)
    assert v.shape == x.shape and v.stride() == (1436, 1) and v.data_ptr() % 16 == 0 and torch.equal(v, x)
    p, d = ops._pad4(v)
    assert d == 1433 and p.shape == (1000, 1436) and p.data_ptr() == v.data_ptr() and torch.equal(p[:, 1433:], torch.zeros(1000, 3, device=dev))
    y = torch.randn(10, 64, device=dev)
    assert ops.align_rows(y) is y


def test_gat_gradients_around_a_hub_target(dev):
    """A 4-layer GAT (8 heads x 16) on 100 nodes around ONE target with 3,617 in-edges: every parameter gradient against the
    oracle computing in float64, within 1e-4 of the gradient's own scale (measured 2e-5: the one-gather backward's per-node
    form of the target-side score gradient holds on a hub). Round 4 blamed that form for the whole-model fuzz's seed 2528
    (first-layer attention gradients 7e-4 off); round 5 traced the seed to a LeakyReLU kink instead, see the next test."""
    from oracle import large as OL
    from rgb_experiment_amd import models as M
    from rgb_experiment_amd.graph import get_graph
    n, f, c, layers, heads = 100, 33, 3, 4, 8
    gen = torch.Generator().manual_seed(2528)
    ei = torch.cat([torch.randint(0, n, (2, 2183), generator=gen),
                    torch.stack([torch.randint(0, n, (3617,), generator=gen), torch.full((3617,), 17)])], dim=1)
    x = torch.randn(n, f, generator=gen)
    y = torch.randint(0, c, (n,), generator=gen)
    mask = torch.rand(n, generator=gen) < 0.6
    torch.manual_seed(7)
    model = M.GAT(num_layers=layers, hidden_unit=16, heads=heads, input_dim=f, output_dim=c, dropout_rate=0.5)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    ref = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    ref = {k: v.requires_grad_(v.is_floating_point() and "running_" not in k) for k, v in ref.items()}
    out = O.gat_forward(ref, x.double(), ei, layers, heads, True)["out"]
    torch.nn.functional.nll_loss(out[mask], y[mask]).backward()
    model.to(dev).train()
    assert get_graph(ei.to(dev), n, 2).fwd.split is not None  # the hub target is a row of the split plan
    loss, _ = model.masked_ce(x.to(dev), ei.to(dev), y.to(dev), mask.to(dev))
    loss.backward()
    got = {k: p.grad.detach().cpu() for k, p in model.named_parameters()}
    rep = OL.compare_grads(got, {k: ref[k].grad.float() for k in got})
    assert rep["max_rel"] < 1e-4, rep


def test_the_hub_seed_of_round_4_is_a_leaky_relu_kink(dev):
    """Whole-model fuzz seed 2528 (4-layer GAT, 100 nodes, a 3,600-in-edge target, an 1,800-out-edge source): convs.0.att_dst's
    gradient 7e-4 of its scale off, whatever the row-split settings and whichever form the target-side pass takes
    (tools/gat_layer_probe.py, profiles/r05_gat_seed2528_kink.txt). Stage by stage every forward activation and every output
    gradient down to layer 1 is closer to float64 than the float32 oracle's; layer 1's backward alone, on float64's inputs, is
    at 1e-6. What differs in the chain: ONE of layer 1's 46,744 pre-activation scores sits 1.2e-6 from zero, float32 rounding
    of the score decides on which side of LeakyReLU's kink it falls, and the DERIVATIVE there is 0.2 or 1. With layer 1's
    att_src scaled by 1 +- 1e-3 (the scores move off the kink, the model is otherwise the same) every gradient of the same
    seed is within 1e-4 of its scale — here pinned; the unmoved seed stays in tools/fuzz_soak.py's range as a known kink."""
    import test_gpu_fuzz as F
    from oracle import large as OL
    from rgb_experiment_amd.graph import clear_cache
    from rgb_experiment_amd.models._stack import masked_ce
    for t in (1e-3, -1e-3):
        desc, model, ref_fn, ei, x, y, masks = F.make_model_case(2528)
        with torch.no_grad():
            model.convs[1].att_src.mul_(1.0 + t)
        sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
        sd = {k: (v.double() if v.is_floating_point() else v.clone()).requires_grad_(v.is_floating_point()) for k, v in sd0.items()}
        out = ref_fn(sd, x.double(), True)["out"]
        torch.nn.functional.nll_loss(out[masks[0]], y[masks[0]]).backward()
        clear_cache()
        model.to(dev).train()
        loss, _ = masked_ce(model, {"x": x.to(dev), "edge_index": ei.to(dev)}, y.to(dev), masks[0].to(dev))
        loss.backward()
        got = {k: p.grad.detach().cpu() for k, p in model.named_parameters()}
        rep = OL.compare_grads(got, {k: sd[k].grad.float() for k in got})
        assert rep["max_rel"] < 1e-4, (t, rep["max_rel"], rep["worst"])


@pytest.mark.parametrize("n,f,out,density", [(3000, 1433, 64, 0.0126), (2708, 1433, 128, 0.0126), (500, 40, 8, 0.05),
                                             (64, 300, 7, 0.0), (1500, 33, 32, 0.09)])
def test_linear_over_the_nonzeros_of_sparse_features(dev, n, f, out, density):
    """ops.prepare_features + ops.linear: x W^T + b and dW, db from the non-zeros of a static feature matrix (two launches of
    the row-gather kernel) against the dense product in float64; empty rows, a dense row and a feature nobody has included."""
    from rgb_experiment_amd import ops
    gen = torch.Generator().manual_seed(n + f)
    x = (torch.rand(n, f, generator=gen) < density).float() * torch.rand(n, f, generator=gen)
    if density:
        x[3] = torch.rand(f, generator=gen)  # one dense row
        x[5] = 0                             # one empty row
        x[:, 2] = 0                          # one feature without a non-zero
    W = torch.randn(out, f, generator=gen, requires_grad=True)
    b = torch.randn(out, generator=gen, requires_grad=True)
    gy = torch.randn(n, out, generator=gen)
    want = x.double() @ W.double().t() + b.double()
    want.backward(gy.double())
    xd = ops.prepare_features(x.to(dev))
    assert getattr(xd, "_rgbx_sparse", None) is not None and torch.equal(xd, x.to(dev))
    Wd, bd = W.detach().to(dev).requires_grad_(True), b.detach().to(dev).requires_grad_(True)
    got = ops.linear(xd, Wd, bd)
    assert type(got.grad_fn).__name__ == "_SparseRowsLinearBackward"
    got.backward(gy.to(dev))
    assert (got.detach().cpu().double() - want.detach()).abs().max().item() < 1e-5 * max(1.0, want.abs().max().item())
    assert (Wd.grad.cpu().double() - W.grad).abs().max().item() < 1e-5 * max(1.0, W.grad.abs().max().item())
    assert (bd.grad.cpu().double() - b.grad).abs().max().item() < 1e-4 * max(1.0, b.grad.abs().max().item())
    dense = torch.randn(50, 16, device=dev)
    assert getattr(ops.prepare_features(dense), "_rgbx_sparse", None) is None  # dense features stay dense


@pytest.mark.parametrize("p", [0.5, 0.1, 1.0])
def test_dropout_on_features_carried_by_their_nonzeros(dev, p):
    """ops.dropout on prepare_features' output in a training forward (reference models/dagnn.py:72): the Bernoulli mask is
    drawn for the non-zeros only; the product and its weight gradient equal the dense product with the SAME dropped matrix
    (rebuilt from the values kept), the kept fraction and the 1 / (1 - p) scale are dropout's, an eval forward and p = 0
    return the features themselves, and a consumer other than this package's Linear cannot read the carrier as a tensor."""
    from rgb_experiment_amd import nn as RN
    from rgb_experiment_amd import ops
    n, f, out = 5000, 1433, 64
    gen = torch.Generator().manual_seed(11)
    x = (torch.rand(n, f, generator=gen) < 0.0126).float() * (0.5 + torch.rand(n, f, generator=gen))
    xd = ops.prepare_features(x.to(dev))
    sp = xd._rgbx_sparse
    assert ops.dropout(xd, p, False) is xd and ops.dropout(xd, 0.0, True) is xd
    torch.manual_seed(3)
    dropped = ops.dropout(xd, p, True)
    assert isinstance(dropped, ops.SparseFeatures) and tuple(dropped.shape) == (n, f)
    sp2 = dropped._rgbx_sparse
    assert sp2.fwd is sp.fwd and sp2.bwd is sp.bwd  # structure shared, values replaced
    kept = sp2.val != 0
    if p < 1.0:
        assert abs(kept.float().mean().item() - (1 - p)) < 0.02
        assert torch.allclose(sp2.val[kept], sp.val[kept] / (1 - p), rtol=1e-6)
    else:
        assert not kept.any()
    assert torch.equal(sp2.val_t, sp2.val[sp.order])
    # the same dropped matrix, dense
    rows = torch.repeat_interleave(torch.arange(n, device=dev), (sp.fwd.rowptr[1:] - sp.fwd.rowptr[:-1]).long())
    dense = torch.zeros(n, f, device=dev)
    dense[rows, sp.fwd.col.long()[:sp.nnz]] = sp2.val[:sp.nnz]
    lin = RN.Linear(f, out).to(dev)
    gy = torch.randn(n, out, generator=gen).to(dev)
    got = lin(dropped)
    got.backward(gy)
    gw = lin.weight.grad.clone()
    lin.weight.grad = None
    want = torch.nn.functional.linear(dense.double(), lin.weight.detach().double(), lin.bias.detach().double())
    assert (got.detach().double() - want).abs().max().item() < 1e-5 * max(1.0, want.abs().max().item())
    want_gw = gy.double().t() @ dense.double()
    assert (gw.double() - want_gw).abs().max().item() < 1e-5 * max(1.0, want_gw.abs().max().item())
    with pytest.raises(Exception):
        torch.relu(dropped)
    # dense features: torch's dropout
    d2 = torch.randn(100, 8, device=dev)
    assert torch.is_tensor(ops.dropout(d2, 0.5, True))


def test_dagnn_trains_on_the_nonzeros_of_bag_of_words_features(dev):
    """DAGNN at the reference's defaults (hidden 64, K = 10, dropout 0.5; models/dagnn.py:68-76) on bag-of-words features:
    the training forward's input dropout and first Linear stay on the non-zeros (no dense [N, F] product, no dense weight
    gradient launch); eval logits equal the float64 oracle's."""
    from oracle import ref_cpu as O
    from rgb_experiment_amd import models as M
    from rgb_experiment_amd import ops
    n, f, c = 3000, 1433, 7
    gen = torch.Generator().manual_seed(5)
    x = torch.zeros(n, f)
    x.scatter_(1, torch.randint(0, f, (n, 18), generator=gen), 1.0)
    x = x / x.sum(1, keepdim=True)
    ei = graph_with_isolated_nodes(n, 20000, 3)
    y = torch.randint(0, c, (n,), generator=gen)
    torch.manual_seed(2)
    model = M.DAGNN(input_dim=f, hidden_dim=64, output_dim=c, K=10, dropout_rate=0.5).to(dev)
    xd = ops.prepare_features(x.to(dev))
    events = []
    ops.set_event_sink(events)
    model.train()
    out = model(xd, ei.to(dev))
    torch.nn.functional.nll_loss(out["out"], y.to(dev)).backward()
    ops.set_event_sink(None)
    kinds = [f"{k}[{getattr(k, 'variant', None)}]" for k, _, _ in events]
    assert any(k.startswith("features_fwd") for k in kinds) and any(k.startswith("features_bwd") for k in kinds), kinds
    assert not any("1433" in k for k in kinds), kinds  # no dense product over the feature width, forward or backward
    assert model.lin1.weight.grad is not None and torch.isfinite(model.lin1.weight.grad).all()
    model.eval()
    with torch.no_grad():
        emb = model(xd, ei.to(dev))["emb"].cpu().double()
    sd = {k: (v.detach().cpu().double() if v.is_floating_point() else v.cpu()) for k, v in model.state_dict().items()}
    ref = O.dagnn_forward(sd, x.double(), ei, 10)["emb"]
    assert (emb - ref).abs().max().item() < 1e-5 * max(1.0, ref.abs().max().item())


def test_gemm_tn_aligns_operands_whose_rows_are_off_the_16_byte_grid(dev):
    """dW = dY^T X with contiguous [K, 7] and [K, 1433] operands (odd widths at their natural stride: what a dropout copy of
    the features or a 7-class logits gradient looks like): ops.gemm_tn makes padded copies and takes the 16-byte path — same
    product as float64, bit-equal to the product of operands aligned beforehand."""
    from rgb_experiment_amd import ops
    K = 20000
    gen = torch.Generator().manual_seed(8)
    a, b = torch.randn(K, 7, generator=gen).to(dev), torch.randn(K, 1433, generator=gen).to(dev)
    got = ops.gemm_tn(a, b)
    want = a.double().t() @ b.double()
    assert (got.double() - want).abs().max().item() < 2e-5 * want.abs().max().item() * (K / 1000) ** 0.5
    assert torch.equal(got, ops.gemm_tn(ops.align_rows(a), ops.align_rows(b)))


def test_experiment_on_bag_of_words_features(dev):
    """experiment() end to end on Cora-shaped data (F = 1433 bag-of-words rows, row-normalised as itexperiments.py:296, C = 7,
    a homophilous graph) for the models whose first Linear meets the features directly: the features are multiplied over
    their non-zeros (ops.prepare_features inside experiment()), DAGNN's input dropout included, with the harness's default
    epoch replay; the runs learn (accuracy far above 1/7) and record every epoch."""
    import rgb_experiment_amd as R
    n, f, c = 2708, 1433, 7
    gen = torch.Generator().manual_seed(21)
    y = torch.randint(0, c, (n,), generator=gen)
    words = torch.randint(0, f, (c, 60), generator=gen)  # every class draws most of its words from its own 60
    pick = torch.where(torch.rand(n, 18, generator=gen) < 0.7, words[y][torch.arange(n).view(-1, 1), torch.randint(0, 60, (n, 18), generator=gen)],
                       torch.randint(0, f, (n, 18), generator=gen))
    x = torch.zeros(n, f)
    x.scatter_(1, pick, 1.0)
    same = (y.view(-1, 1) == y.view(1, -1)) & (torch.rand(n, n, generator=gen) < 0.002)
    ei = same.nonzero().t().contiguous()
    data = R.Data(x=x, y=y, edge_index=ei)
    seen = []
    real_prepare = R.ops.prepare_features
    try:
        R.ops.prepare_features = lambda t, *a, **k: seen.append(real_prepare(t, *a, **k)) or seen[-1]
        for name, params in (("DAGNN", R.InitialParameters.defaults_for("dagnn")), ("GCN", R.InitialParameters.defaults_for("gcn")),
                             ("GraphSAGE2", R.InitialParameters.defaults_for("graphsage2"))):
            res = R.experiment(params, specify_data=True, data=data, model_name=name, learning_rate=0.01, epoch=30,
                               need_to_reappear=True, print_print=False, return_model=True)
            assert res["ACC"] > 0.5, (name, res["ACC"])
            assert len(res["history"]["val_acc"]) == 30
    finally:
        R.ops.prepare_features = real_prepare
    assert len(seen) == 3 and all(getattr(t, "_rgbx_sparse", None) is not None for t in seen)


@pytest.mark.parametrize("d", [4, 8, 16, 40, 64, 128, 256])
@pytest.mark.parametrize("weighted", [True, False])
def test_short_rows_gather_matches_the_row_per_wave_kernel(dev, d, weighted):
    """rgbx_spmm_csr_short_rows_f32 (a lane group per target row, slot order) against float64 and against rgbx_spmm_csr_f32
    on the same CSR: empty rows, rows of one slot, one row of 3,000 slots, a row count that fills no whole wave, a bias;
    reproducible; refuses widths it has no form for."""
    from rgb_experiment_amd import _lib, ops
    from rgb_experiment_amd.graph import CSR
    n, n_src = 3001, 700
    gen = torch.Generator().manual_seed(d)
    lens = torch.randint(0, 40, (n,), generator=gen)
    lens[5], lens[6], lens[17] = 0, 1, 3000
    rowptr = torch.zeros(n + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(lens, 0)
    nnz = int(rowptr[-1])
    col = torch.randint(0, n_src, (nnz,), generator=gen)
    w = torch.rand(nnz, generator=gen) + 0.1 if weighted else None
    x = torch.randn(n_src, d, generator=gen)
    bias = torch.randn(d, generator=gen)
    csr = CSR(rowptr.to(torch.int32).to(dev), col.to(torch.int32).to(dev), None, n, nnz, None)
    wd = None if w is None else w.to(dev)
    got = ops.spmm_short_rows_raw(csr, wd, x.to(dev), bias=bias.to(dev))
    rows = torch.repeat_interleave(torch.arange(n), lens)
    vals = x.double()[col] * (w.double().view(-1, 1) if weighted else 1.0)
    want = torch.zeros(n, d, dtype=torch.float64).index_add_(0, rows, vals) + bias.double()
    assert (got.cpu().double() - want).abs().max().item() < 1e-5 * max(1.0, want.abs().max().item())
    other = ops.spmm_raw(csr, wd, None, x.to(dev), bias=bias.to(dev))
    assert (got - other).abs().max().item() < 1e-4 * max(1.0, want.abs().max().item())
    assert torch.equal(got, ops.spmm_short_rows_raw(csr, wd, x.to(dev), bias=bias.to(dev)))
    assert ops.short_rows_ok(csr, x.to(dev)) == (d <= 64)  # wider: both forms run at L2's rate, the row-per-wave one stays
    assert not ops.short_rows_ok(csr, torch.empty(700, 6, device=dev))       # width not a multiple of 4
    assert not ops.short_rows_ok(csr, torch.empty(200000, 64, device=dev))   # a table beyond the cache budget
    with pytest.raises(RuntimeError):
        ops.spmm_short_rows_raw(csr, wd, torch.randn(n_src, 6, device=dev))


@pytest.mark.parametrize("name", ["gcn", "graphsage", "graphsage2"])
@pytest.mark.parametrize("layers,f,hid,c", [(2, 1433, 64, 7), (3, 300, 64, 3), (4, 128, 32, 7), (2, 128, 64, 40)])
def test_eval_forward_collapsed_to_the_class_width(dev, name, layers, f, hid, c):
    """ConvStack._run_collapsed: the reference's stacks have no activation between their layers (models/gcn.py:25-31,
    graphsage.py:26-32, graphsage2.py:27-33), so an eval forward is a polynomial in the aggregation operator, evaluated by
    Horner's rule with every gather at the class width: GCN A_hat(... A_hat(X P) + c ...) + b, the SAGE stacks
    U = A(U) + X Q_k + d_k with one product X [Q_0 | .. | Q_L]. Eval logits, one-mask and two-mask statistics against the
    float64 oracle and against the layer-by-layer route (collapse_eval = False), after a training step has moved the running
    statistics; isolated targets (SAGEConv aggregates 0 there) and a hub row; every aggregation launch of the collapsed
    forward at width pad4(C); wide-class models keep the layer-by-layer route."""
    from oracle import ref_cpu as O
    from rgb_experiment_amd import models as M
    from rgb_experiment_amd import ops
    from rgb_experiment_amd.models._stack import masked_ce, masked_ce_pair
    n = 3001
    gen = torch.Generator().manual_seed(layers * 1000 + c)
    x = torch.randn(n, f, generator=gen)
    ei = graph_with_isolated_nodes(n, 24000, c, hub=1500)
    y = torch.randint(0, c, (n,), generator=gen)
    r = torch.rand(n, generator=gen)
    m_a, m_b = r < 0.3, r > 0.6
    torch.manual_seed(7)
    cls = {"gcn": M.GCN, "graphsage": M.GraphSAGE, "graphsage2": M.GraphSAGE2}[name]
    oracle_forward = {"gcn": O.gcn_forward, "graphsage": O.graphsage_forward, "graphsage2": O.graphsage2_forward}[name]
    model = cls(num_layers=layers, hidden_unit=hid, input_dim=f, output_dim=c, dropout_rate=0.5).to(dev)
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() == 1:
                p.uniform_(0.5, 1.5) if p.mean().item() > 0.9 else p.uniform_(-0.5, 0.5)
    xd, eid, yd = x.to(dev), ei.to(dev), y.to(dev)
    model.train()
    masked_ce(model, {"x": xd, "edge_index": eid}, yd, m_a.to(dev))[0].backward()  # running statistics move off their init
    model.eval()
    sd = {k: (v.detach().cpu().double() if v.is_floating_point() else v.cpu()) for k, v in model.state_dict().items()}
    ref = oracle_forward(sd, x.double(), ei, layers, False)
    want = torch.stack([torch.stack([torch.nn.functional.nll_loss(ref["out"][m], y[m], reduction="sum"),
                                     m.sum().double(), (ref["out"][m].argmax(1) == y[m]).sum().double()]) for m in (m_a, m_b)])
    events = []
    with torch.no_grad():
        assert model._collapsed_operands() is not None
        ops.set_event_sink(events)
        emb = model(xd, eid)["emb"]
        pair = masked_ce_pair(model, {"x": xd, "edge_index": eid}, yd, m_a.to(dev), m_b.to(dev))
        _, one = masked_ce(model, {"x": xd, "edge_index": eid}, yd, m_a.to(dev))
        ops.set_event_sink(None)
        model.collapse_eval = False
        emb_layers = model(xd, eid)["emb"]
        pair_layers = masked_ce_pair(model, {"x": xd, "edge_index": eid}, yd, m_a.to(dev), m_b.to(dev))
        model.collapse_eval = True
    scale = max(1.0, ref["emb"].abs().max().item())
    assert tuple(emb.shape) == (n, c)
    assert (emb.cpu().double() - ref["emb"]).abs().max().item() < 2e-5 * scale
    assert (emb - emb_layers).abs().max().item() < 4e-5 * scale
    for got in (pair.cpu(), pair_layers.cpu()):
        assert torch.equal(got[:, 1], want[:, 1])
        assert (got[:, 0] - want[:, 0]).abs().max().item() < 1e-4 * max(1.0, want[:, 0].abs().max().item())
        assert (got[:, 2] - want[:, 2]).abs().max().item() <= 2
    assert torch.equal(one.cpu()[1], want[0, 1]) and abs(one.cpu()[0].item() - want[0, 0].item()) < 1e-4 * max(1.0, want[0, 0].item())
    widths = [int(str(getattr(k, "variant", "")).split("+d")[-1]) for k, _, _ in events
              if str(k).startswith(("gcn_", "mean_")) and "+d" in str(getattr(k, "variant", ""))]
    assert len(widths) == 3 * layers and set(widths) == {(c + 3) // 4 * 4}, widths
    # a model whose classes are as wide as its hidden layers is not collapsed
    wide = cls(num_layers=2, hidden_unit=32, input_dim=f, output_dim=40, dropout_rate=0.5).to(dev).eval()
    assert wide._collapsed_operands() is None
