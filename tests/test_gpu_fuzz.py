"""Seeded differential fuzz of the conv layers against the oracle: random layer kind, graph shape (single nodes,
edgeless graphs, rows without in-edges, self-loops, duplicates, a hub), widths on both sides of every dispatch
boundary (odd, < 4, not a multiple of 32, in > out, in <= out, > 128), with and without an input gradient, train
and no_grad. Every case checks the output and every gradient. The parity tests proper pin chosen shapes; this walks
the space between them."""
import random

import pytest
import torch

from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu

WIDTHS = [1, 2, 3, 4, 5, 7, 8, 12, 16, 31, 32, 33, 40, 64, 96, 100, 128, 130, 160, 256]
KINDS = ["gcn", "sage", "mysage", "gat", "appnp", "gin", "sgc", "dagnn"]


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda")


def make_graph(rng, n):
    g = torch.Generator().manual_seed(rng.randrange(1 << 30))
    style = rng.choice(["empty", "sparse", "dense", "hub", "loops"])
    if style == "empty" or n == 0:
        return torch.zeros((2, 0), dtype=torch.int64)
    e = {"sparse": n // 2 + 1, "dense": 12 * n, "hub": 4 * n, "loops": 3 * n}[style]
    ei = torch.randint(0, n, (2, e), generator=g)
    if style == "hub":  # one target with most of the edges, one source likewise
        k = 6 * n + 3000
        ei = torch.cat([ei, torch.stack([torch.randint(0, n, (k,), generator=g), torch.full((k,), rng.randrange(n))]),
                        torch.stack([torch.full((k // 2,), rng.randrange(n)), torch.randint(0, n, (k // 2,), generator=g)])],
                       dim=1)
    if style == "loops":
        s = torch.randint(0, n, (n,), generator=g)
        ei = torch.cat([ei, torch.stack([s, s]), ei[:, : max(1, e // 3)]], dim=1)
    return ei[:, torch.randperm(ei.size(1), generator=g)]


def tol(ref, base=2e-4):
    return base * max(1.0, float(ref.detach().abs().max()) if ref.numel() else 1.0)


def check(name, got, ref, base=2e-4):
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    if ref.numel():
        err = (got.detach().cpu().to(ref.dtype) - ref.detach()).abs().max().item()
        assert err < tol(ref, base), (name, err, tol(ref, base))


def run_case(dev, seed, ref_dtype=torch.float32):
    """`ref_dtype=torch.float64`: the oracle computes the same case in double (same float32 inputs and weights) — the
    adjudication for a seed whose float32 oracle and HIP results differ by more than the tolerance (tools/fuzz_soak.py): long
    rows of duplicate edges are summed sequentially by the float32 oracle (index_add_), which is then the less accurate side."""
    from rgb_experiment_amd import nn as RN
    from rgb_experiment_amd import ops
    from rgb_experiment_amd.graph import LOOPS_ADD_REMAINING, clear_cache, get_graph
    rng = random.Random(seed)
    kind = KINDS[seed % len(KINDS)]
    n = rng.choice([1, 2, 3, 31, 32, 33, 64, 100, 257, 700])
    f_in, f_out = rng.choice(WIDTHS), rng.choice(WIDTHS)
    ei = make_graph(rng, n)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, f_in, generator=g)
    need_xgrad = rng.random() < 0.6
    no_grad = rng.random() < 0.15
    torch.manual_seed(seed)
    desc = f"seed={seed} kind={kind} n={n} E={ei.size(1)} in={f_in} out={f_out} xgrad={need_xgrad} no_grad={no_grad}"

    if kind == "gcn":
        conv = RN.GCNConv(f_in, f_out)
        ref_fn = lambda sd, xc: O.gcn_conv(xc, ei, sd["lin.weight"], sd["bias"])
    elif kind == "sage":
        conv = RN.SAGEConv(f_in, f_out)
        ref_fn = lambda sd, xc: O.sage_conv(xc, ei, sd["lin_l.weight"], sd["lin_l.bias"], sd["lin_r.weight"])
    elif kind == "mysage":
        conv = RN.MySAGEConv(f_in, f_out)
        ref_fn = lambda sd, xc: O.my_sage_conv(xc, ei, sd["lin_l.weight"], sd["lin_l.bias"], sd["lin_r.weight"],
                                               sd["lin_r.bias"])
    elif kind == "gat":
        H = rng.choice([1, 2, 3, 8])
        C = rng.choice([1, 2, 5, 8, 16, 64, 128]) if H == 1 else rng.choice([1, 4, 7, 16, 32])
        concat = rng.random() < 0.7
        conv = RN.GATConv(f_in, C, H, concat=concat)
        desc += f" H={H} C={C} concat={concat}"
        ref_fn = lambda sd, xc: O.gat_conv(xc, ei, sd["lin_src.weight"], sd["att_src"], sd["att_dst"], sd["bias"], H,
                                           concat)
    elif kind == "appnp":
        K, alpha = rng.choice([0, 1, 2, 5]), rng.choice([0.0, 0.1, 0.5])
        conv = RN.APPNP(K, alpha)
        desc += f" K={K} alpha={alpha}"
        ref_fn = lambda sd, xc: O.appnp(xc, ei, K, alpha)
    elif kind == "gin":
        hid = rng.choice([8, 32, 64])
        conv = RN.GINConv(torch.nn.Sequential(RN.Linear(f_in, hid), torch.nn.ReLU(), RN.Linear(hid, f_out)),
                          train_eps=True)
        with torch.no_grad():
            conv.eps.fill_(rng.choice([0.0, 0.25, -0.5]))

        def ref_fn(sd, xc):
            agg = O.propagate(ei, xc, n, None, "add") + (1 + sd["eps"]) * xc
            h = torch.relu(agg @ sd["nn.0.weight"].t() + sd["nn.0.bias"])
            return h @ sd["nn.2.weight"].t() + sd["nn.2.bias"]
    elif kind == "sgc":
        K = rng.choice([1, 2, 3])
        conv = RN.SGConv(f_in, f_out, K=K)
        desc += f" K={K}"

        def ref_fn(sd, xc):
            e2, w = O.gcn_norm(ei, None, n)
            h = xc
            for _ in range(K):
                h = O.propagate(e2, h, n, w, "add")
            return h @ sd["lin.weight"].t() + sd["lin.bias"]
    else:  # dagnn's Prop through ops.dagnn_prop
        K = rng.choice([0, 1, 3, 6])
        conv = torch.nn.Linear(f_in, 1)  # proj
        desc += f" K={K}"

        def ref_fn(sd, xc):
            e2, w = O.gcn_norm(ei, None, n)
            preds, h = [xc], xc
            for _ in range(K):
                h = O.propagate(e2, h, n, w, "add")
                preds.append(h)
            pps = torch.stack(preds, dim=1)
            retain = torch.sigmoid((pps @ sd["weight"].t() + sd["bias"]).squeeze(-1))
            return torch.matmul(retain.unsqueeze(1), pps).squeeze(1)

    with torch.no_grad():  # biases start at zero in several layers: make them count
        for p in conv.parameters():
            if p.dim() == 1 and p.numel() > 1:
                p.uniform_(-0.5, 0.5)
    sd = {k: (v.detach().clone().to(ref_dtype) if v.is_floating_point() else v.detach().clone())
          .requires_grad_(v.is_floating_point()) for k, v in conv.state_dict().items() if "lin_dst" not in k}
    conv.to(dev)
    clear_cache()
    xd = x.to(dev).requires_grad_(need_xgrad and not no_grad)
    xc = x.clone().to(ref_dtype).requires_grad_(need_xgrad and not no_grad)
    eid = ei.to(dev)

    def fwd():
        if kind == "dagnn":
            if f_in > 256:
                return None
            return ops.dagnn_prop(xd, get_graph(eid, n, LOOPS_ADD_REMAINING), K, conv.weight, conv.bias)
        return conv(xd, eid)

    try:
        if no_grad:
            with torch.no_grad():
                got = fwd()
        else:
            got = fwd()
        ref = ref_fn(sd, xc)
        check("out", got, ref)
        if no_grad or not ref.requires_grad:
            return
        go = torch.randn(ref.shape, generator=g)
        got.backward(go.to(dev))
        ref.backward(go.to(ref_dtype))
        if need_xgrad:
            check("x.grad", xd.grad, xc.grad)
        for name, p in conv.named_parameters():
            if "lin_dst" in name or name not in sd:
                continue
            rg = sd[name].grad
            if rg is None:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
                continue
            assert p.grad is not None, name
            check(name + ".grad", p.grad, rg, 4e-4)
    except AssertionError as exc:
        raise AssertionError(f"{desc}: {exc}") from exc
    except RuntimeError as exc:
        raise RuntimeError(f"{desc}: {exc}") from exc


# Seeds in the suite (round 5: 154 of the 795 seeds rounds 3-4 pinned — the first 8 / 10 / 15 of every block, so every kind
# and every block's shape family stays; the suite had 590 of its 900 s): the others, and thousands beyond, are
# tools/fuzz_soak.py's (profiles/r05_fuzz_soak.txt; `python tools/fuzz_soak.py 9000 9320`, `--models 0 75`, `--fused 0 400`).
@pytest.mark.parametrize("block", range(8))
def test_conv_layers_against_the_oracle_on_random_shapes(dev, block):
    for seed in range(block * 40, block * 40 + 8):
        run_case(dev, 9000 + seed)


# ---- whole models through the routes experiment() takes: loss inside the last conv's kernel, BatchNorm handed to the
# following conv, column statistics from the producing launch, two statistics sets from one eval forward -----------

MODEL_KINDS = ["gcn", "graphsage", "graphsage2", "gat", "appnpstack"]


def _close(name, got, ref64, ref32, scale_floor=0.0, rel=3e-4, own_factor=16.0):
    """|HIP - float64 oracle| within rel x the reference's scale — or within own_factor x what the float32 ORACLE itself is off
    from the float64 one: a badly conditioned case (BatchNorm over a handful of rows, rank-one inputs, thousands of duplicate edges
    between two nodes) moves every float32 implementation, and the bound then follows the case instead of failing it."""
    got = got.detach().cpu().double()
    err = (got - ref64).abs().max().item() if ref64.numel() else 0.0
    scale = max(ref64.abs().max().item() if ref64.numel() else 0.0, scale_floor, 1e-30)
    own = (ref32.double() - ref64).abs().max().item() if ref64.numel() else 0.0
    bound = max(rel * scale, own_factor * own)
    assert err <= bound, (name, err, bound, "float32 oracle off by", own)


def make_model_case(seed):
    """(description, model on the CPU, oracle forward (sd, x, training) -> dict, edge_index, x, y, masks) of a seed."""
    from rgb_experiment_amd import models as M
    rng = random.Random(seed)
    kind = MODEL_KINDS[seed % len(MODEL_KINDS)]
    n = rng.choice([2, 3, 33, 64, 100, 257, 700, 2000])
    f_in = rng.choice([1, 3, 8, 12, 32, 33, 64, 100, 128, 160])
    classes = rng.choice([2, 3, 7, 32, 40, 128, 130])
    hidden = rng.choice([4, 16, 32, 64, 96, 100, 128])
    layers = rng.choice([2, 2, 3, 4])
    ei = make_graph(rng, n)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, f_in, generator=g)
    y = torch.randint(0, classes, (n,), generator=g)
    r = torch.rand(n, generator=g)
    masks = [r < 0.5, (r >= 0.5) & (r < 0.75), r >= 0.75]
    masks[0][rng.randrange(n)] = True  # at least one training row
    desc = f"seed={seed} model={kind} n={n} E={ei.size(1)} in={f_in} hidden={hidden} classes={classes} layers={layers}"
    if seed >= 6000 and seed % 3 == 0 and n >= 33 and f_in >= 8:  # (smaller: 8 % of nothing is an all-zero matrix)
        # bag-of-words-like features (round 5; seeds below 6000 keep the cases earlier soaks ran): about 8 % of the entries
        # non-zero, one all-zero row — run_model_case hands them to ops.prepare_features, so the first Linear runs over the
        # non-zeros (row-gather / short-rows kernels, transposed gather for its weight gradient)
        keep = torch.rand(n, f_in, generator=torch.Generator().manual_seed(seed + 1_000_000)) < 0.08
        keep[0] = False
        x = x.abs() * keep
        desc += " sparse-features"
    torch.manual_seed(seed)
    if kind == "gcn":
        model = M.GCN(num_layers=layers, hidden_unit=hidden, input_dim=f_in, output_dim=classes, dropout_rate=0.5)
        ref_fn = lambda sd, xc, tr: O.gcn_forward(sd, xc, ei, layers, tr)
    elif kind == "graphsage":
        model = M.GraphSAGE(num_layers=layers, hidden_unit=hidden, input_dim=f_in, output_dim=classes, dropout_rate=0.5)
        ref_fn = lambda sd, xc, tr: O.graphsage_forward(sd, xc, ei, layers, tr)
    elif kind == "graphsage2":
        model = M.GraphSAGE2(num_layers=layers, hidden_unit=hidden, input_dim=f_in, output_dim=classes, dropout_rate=0.5)
        ref_fn = lambda sd, xc, tr: O.graphsage2_forward(sd, xc, ei, layers, tr)
    elif kind == "gat":
        heads = rng.choice([1, 2, 8])
        hidden = rng.choice([1, 4, 8, 16])
        desc += f" heads={heads} per-head={hidden}"
        model = M.GAT(num_layers=layers, hidden_unit=hidden, heads=heads, input_dim=f_in, output_dim=classes, dropout_rate=0.5)
        ref_fn = lambda sd, xc, tr: O.gat_forward(sd, xc, ei, layers, heads, tr)
    else:
        K, alpha = rng.choice([1, 2, 10]), rng.choice([0.1, 0.5])
        desc += f" K={K} alpha={alpha}"
        model = M.APPNPStack(hidden_unit=hidden, input_dim=f_in, output_dim=classes, K=K, alpha=alpha, dropout_rate=0.5)
        ref_fn = lambda sd, xc, tr: O.appnp_stack_forward(sd, xc, ei, K, alpha, tr)
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() == 1:  # biases start at zero, BatchNorm weights at one: make them count
                p.uniform_(0.5, 1.5) if p.mean().item() > 0.9 else p.uniform_(-0.5, 0.5)
    return desc, model, ref_fn, ei, x, y, masks


def run_model_case(dev, seed):
    """One training step (loss + every parameter gradient) and one eval forward (val + test statistics from ONE forward, logits)
    of a random model on a random graph, HIP vs the oracle under autograd — the oracle in float64 as the reference and in
    float32 as the yardstick of the case's conditioning (_close). Gradients: 3e-4 of the parameter's largest reference entry,
    floored at 1 % of the model's largest gradient entry (a bias in front of a BatchNorm has a TRUE gradient of zero)."""
    from rgb_experiment_amd.graph import clear_cache
    from rgb_experiment_amd.models._stack import masked_ce, masked_ce_pair
    desc, model, ref_fn, ei, x, y, masks = make_model_case(seed)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    nll = torch.nn.functional.nll_loss
    # BatchNorm over two or three rows, or over rank-one features (one input column: every hidden column is the same
    # normalised vector up to sign, and what reaches the first layers' parameters is the rounding left of a projection that
    # cancels): such cases stay in — they are where shapes degenerate — but only as a detector of gross errors
    # (tools/fuzz_soak.py --models 0 1500, round 4: 60 of 67 exceedances at 2e-4 had n <= 3, 6 of the other 7 one input column)
    conditioned = x.size(0) >= 33 and x.size(1) >= 3
    tol = dict(rel=3e-4, own_factor=16.0) if conditioned else dict(rel=5e-2, own_factor=64.0)

    def oracle_step(dtype):
        sd = {k: (v.detach().clone().to(dtype) if v.is_floating_point() else v.clone()).requires_grad_(v.is_floating_point())
              for k, v in sd0.items()}
        ref = ref_fn(sd, x.to(dtype), True)
        loss = nll(ref["out"][masks[0]], y[masks[0]])
        loss.backward()
        return loss.detach(), {k: v.grad.detach() for k, v in sd.items() if v.is_floating_point() and v.grad is not None}

    def oracle_eval(sd1, dtype):
        sd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd1.items()}
        ref = ref_fn(sd, x.to(dtype), False)
        sums = [nll(ref["out"][m], y[m], reduction="sum") if int(m.sum()) else torch.zeros((), dtype=dtype) for m in masks[1:]]
        hits = [int((ref["out"][m].argmax(1) == y[m]).sum()) for m in masks[1:]]
        return ref["emb"].detach(), torch.stack(sums).detach(), hits

    model.to(dev)
    clear_cache()
    xd, eid, yd = x.to(dev), ei.to(dev), y.to(dev)
    if "sparse-features" in desc:
        from rgb_experiment_amd import ops
        xd = ops.prepare_features(xd)
        assert getattr(xd, "_rgbx_sparse", None) is not None or int((x != 0).sum()) > 0.1 * x.numel()
    md = [m.to(dev) for m in masks]
    try:
        model.train()
        loss, stats = masked_ce(model, {"x": xd, "edge_index": eid}, yd, md[0])
        loss.backward()
        (l32, g32), (l64, g64) = oracle_step(torch.float32), oracle_step(torch.float64)
        _close("loss", loss, l64, l32, scale_floor=1.0, rel=tol["rel"] / 10, own_factor=tol["own_factor"])
        assert int(stats[1].item()) == int(masks[0].sum())
        floor = 0.01 * max(v.abs().max().item() for v in g64.values())
        for name, p in model.named_parameters():
            if name in g64:
                assert p.grad is not None, name
                _close(name + ".grad", p.grad, g64[name], g32[name], scale_floor=floor, **tol)
        # eval forward on the model as the training forward left it (running statistics moved): both masks from one forward
        sd1 = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        model.eval()
        with torch.no_grad():
            pair = masked_ce_pair(model, {"x": xd, "edge_index": eid}, yd, md[1], md[2]).cpu()
            emb = model(xd, eid)["emb"]
        (e32, s32, _), (e64, s64, hits) = oracle_eval(sd1, torch.float32), oracle_eval(sd1, torch.float64)
        _close("eval logits", emb, e64, e32, scale_floor=1.0, **tol)
        _close("eval nll sums", pair[:, 0], s64, s32, scale_floor=1.0, **tol)
        for i, m in enumerate(masks[1:]):
            cnt = int(m.sum())
            assert int(pair[i, 1].item()) == cnt, ("count", i)
            assert abs(int(pair[i, 2].item()) - hits[i]) <= max(1, cnt // 200), ("hits", i, pair[i, 2].item(), hits[i])
    except AssertionError as exc:
        raise AssertionError(f"{desc}: {exc}") from exc
    except RuntimeError as exc:
        raise RuntimeError(f"{desc}: {exc}") from exc


@pytest.mark.parametrize("block", range(3))
def test_whole_models_against_the_oracle_on_random_shapes(dev, block):
    for seed in range(block * 25, block * 25 + 10):  # tools/fuzz_soak.py --models soaked 3,500
        run_model_case(dev, seed)


def test_whole_models_on_sparse_features(dev):
    """Seeds 6000 .. 6089, every third with bag-of-words-like features multiplied over their non-zeros (make_model_case)."""
    for seed in range(6000, 6090, 3):
        run_model_case(dev, seed)


@pytest.mark.parametrize("seed", [557])
def test_whole_model_seeds_the_soaks_found(dev, seed):
    """557 (tools/fuzz_soak.py --models 75 1500, round 5): graphsage2 on 2 nodes, hidden 100 — BatchNorm statistics handed over
    by the row kernel's epilogue on columns with std 4e-4 under a mean of 1 (tests/test_gpu_rows.py
    test_column_statistics_of_nearly_constant_columns)."""
    run_model_case(dev, seed)


# ---- rgbx_fused_layer_f32 by option: dense or aggregating, blocked or plain rows, pre-affine, root term, stored z, column
# sums, blocked output, loss statistics (one or two masks) or loss gradient -----------------------------------------------

def _to_blocked(m, B):
    n, d = m.shape
    return m.view(n, B, d // B).permute(1, 0, 2).contiguous()


def run_fused_layer_case(dev, seed):
    from rgb_experiment_amd import ops
    from rgb_experiment_amd.graph import LOOPS_ADD_REMAINING, clear_cache, get_graph
    rng = random.Random(seed)
    n = rng.choice([1, 2, 31, 32, 33, 64, 97, 257, 1013, 4100])
    K = rng.choice([4, 8, 12, 32, 48, 64, 96, 128, 160, 256])
    n_out = rng.choice([32, 64, 96, 128, 160, 256])
    dense = rng.random() < 0.6
    g = torch.Generator().manual_seed(seed)
    d = lambda t: t.to(dev)
    x = torch.randn(n, K, generator=g)
    W = torch.randn(n_out, K, generator=g) / K ** 0.5
    bias = torch.randn(n_out, generator=g) if rng.random() < 0.8 else None
    pre = None
    if rng.random() < 0.5:
        pre = (torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g), torch.rand(n, generator=g))
    root = rng.random() < 0.4
    xr = torch.randn(n, K, generator=g) if root else None
    Wr = torch.randn(n_out, K, generator=g) / K ** 0.5 if root else None
    in_blocks = [b for b in (1, 2, 4, 8) if K % b == 0 and (K // b) % 4 == 0]
    B = rng.choice(in_blocks) if dense else 1
    Br = rng.choice(in_blocks)
    out_blocks = [b for b in (1, 2, 3, 4, 5, 8) if n_out % b == 0 and (n_out // b) % 8 == 0]
    Bo = rng.choice(out_blocks)
    mode = rng.choice(["out", "out+z", "out+colsums", "out+blocked", "blocked only", "ce stats", "ce two masks", "ce grad"])
    if mode.startswith("ce") and n_out > 128:
        mode = "out+z"
    desc = (f"seed={seed} n={n} K={K} Nout={n_out} {'dense' if dense else 'aggregating'} B={B} Br={Br if root else '-'} "
            f"Bo={Bo} pre={pre is not None} root={root} bias={bias is not None} mode={mode}")
    # reference in float64
    x64 = x.double()
    if dense:
        z = x64
        csr = w = None
        rowsum_ref = None
    else:
        ei = make_graph(rng, n)
        clear_cache()
        gr = get_graph(ei.to(dev), n, LOOPS_ADD_REMAINING)
        csr, w = gr.fwd, gr.w
        e2, wn = O.gcn_norm(ei, None, n)
        z = O.propagate(e2, x64, n, wn.double(), "add")
        if pre is not None:  # the affine map of the aggregate needs the row sums of the operator (ops: graph.rowsum)
            pre = (pre[0], pre[1], torch.zeros(n).index_add_(0, e2[1], wn))
    if pre is not None:
        z = z * pre[0].double() + pre[1].double() * pre[2].double()[:, None]
    out = z @ W.double().t()
    if bias is not None:
        out = out + bias.double()
    if root:
        r = xr.double()
        if pre is not None:
            r = r * pre[0].double() + pre[1].double()
        out = out + r @ Wr.double().t()
    try:
        kw = dict(bias=None if bias is None else d(bias), pre=None if pre is None else tuple(d(t) for t in pre))
        if not dense:
            kw.update(csr=csr, w=w)
        if root:
            kw.update(x_root=_to_blocked(d(xr), Br) if (Br > 1 and (dense or not mode.startswith("ce"))) else d(xr),
                      wt_root=d(Wr).t().contiguous())
        xin = _to_blocked(d(x), B) if B > 1 else d(x)
        wt = d(W).t().contiguous()
        tol_out = 2e-4 * max(1.0, out.abs().max().item())
        if mode.startswith("ce"):
            y = torch.randint(0, n_out, (n,), generator=g)
            if n > 2:
                y[rng.randrange(n)] = -1  # an ignored row
            ma, mb = torch.rand(n, generator=g) < 0.6, torch.rand(n, generator=g) < 0.3
            logp = torch.log_softmax(out, 1)

            def want_stats(m):
                sel = m & (y >= 0)
                nll = -logp[sel, y[sel]].sum().item() if int(sel.sum()) else 0.0
                return nll, int(sel.sum()), int((out[sel].argmax(1) == y[sel]).sum())
            if mode == "ce two masks":
                _, _, st = ops.fused_layer(xin, wt, ce=(d(y), (d(ma), d(mb)), None), **kw)
                for i, m in enumerate((ma, mb)):
                    nll, cnt, hits = want_stats(m)
                    assert int(st[i, 1].item()) == cnt and abs(st[i, 0].item() - nll) < 2e-4 * max(1.0, abs(nll)), (i, st[i], nll, cnt)
                    assert abs(int(st[i, 2].item()) - hits) <= max(1, cnt // 100), ("hits", i)
            else:
                nll, cnt, hits = want_stats(ma)
                scale = ops.mask_scale(d(y), d(ma), n_out) if (mode == "ce grad" and cnt) else None
                dl, _, st = ops.fused_layer(xin, wt, ce=(d(y), d(ma), scale), **kw)
                assert int(st[1].item()) == cnt and abs(st[0].item() - nll) < 2e-4 * max(1.0, abs(nll)), (st, nll, cnt)
                if scale is not None:
                    sel = ma & (y >= 0)
                    want = torch.softmax(out, 1)
                    want[torch.arange(n), y.clamp(min=0)] -= 1.0
                    want = want * sel[:, None] / cnt
                    assert (dl.cpu().double() - want).abs().max().item() < 1e-6 + 2e-4 * want.abs().max().item(), "loss gradient"
            return
        ob = torch.empty((Bo, n, n_out // Bo), device=dev) if "blocked" in mode else None
        got, zz, cs = ops.fused_layer(xin, wt, want_out=mode != "blocked only", out_blocked=ob, want_z=mode == "out+z",
                                      want_colsums=mode == "out+colsums", **kw)
        if mode != "blocked only":
            assert (got.cpu().double() - out).abs().max().item() < tol_out, "out"
        if ob is not None:
            rows = ob.permute(1, 0, 2).reshape(n, n_out)
            assert (rows.cpu().double() - out).abs().max().item() < tol_out, "blocked out"
            if got is not None:
                assert torch.equal(rows, got), "blocked copy differs from the rows"
        if mode == "out+z":
            assert (zz.cpu().double() - z).abs().max().item() < 2e-5 * max(1.0, z.abs().max().item()), "z"
        if mode == "out+colsums":
            want_cs = torch.stack([out.sum(0), (out ** 2).sum(0)])
            assert (cs.cpu() - want_cs).abs().max().item() < 2e-4 * max(1.0, want_cs.abs().max().item()), "colsums"
    except AssertionError as exc:
        raise AssertionError(f"{desc}: {exc}") from exc
    except RuntimeError as exc:
        raise RuntimeError(f"{desc}: {exc}") from exc


@pytest.mark.parametrize("block", range(4))
def test_fused_layer_forms_on_random_shapes(dev, block):
    for seed in range(block * 100, block * 100 + 15):
        run_fused_layer_case(dev, seed)
