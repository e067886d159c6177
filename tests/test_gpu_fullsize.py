"""Parity at the BASELINE sizes (S: |V|=200k |E|=4M, L: |V|=2M |E|=60M, d=128), on the kernels the benchmark times.

The PyG-dataflow oracle cannot hold these graphs ([E', d] temporaries of 32 GB), so:
  * the fused aggregate+transform kernel (rgbx_spmm_linear_f32, what bench.py's timed region launches) is
    compared on the WHOLE graph with the C restatement's propagate (oracle/propagate_ref.c, per-target sums)
    followed by a CPU matmul;
  * whole-model logits of every BASELINE config (gcn / graphsage / graphsage2 / gat / appnpstack) are compared at
    sampled target rows with the UNCHANGED oracle forward on the targets' in-neighbourhood (oracle/sampled.py);
  * APPNP at its stated K = 10 (whose ten hops cover the whole graph, so no neighbourhood can be cut out) is compared
    on ALL rows with ten iterations of the C restatement.
  * BACKWARD (reference itexperiments.py:439): one train-mode forward + backward of every BASELINE model, every
    parameter's gradient — at S against the full oracle under torch autograd (its [E', d] temporaries fit the host
    there), at L against oracle/large.py (propagate and its adjoint through the C restatement over the CSR and the
    transposed CSR, dense layers / BatchNorm / loss under CPU autograd; pinned to ref_cpu by tests/test_oracle_large.py).
Tolerance: 1e-4 absolute on logits (north_star); gradients 1e-4 * max(1, |g|_inf) AND 2e-3 relative to |g|_inf."""
import pytest
import torch

from oracle import large as OL
from oracle import ref_cpu as O
from oracle import sampled as S

pytestmark = pytest.mark.gpu

TOL = 1e-4
SIZES = {"S": (200_000, 4_000_000), "L": (2_000_000, 60_000_000)}
_CACHE = {}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def workload(size):
    """bench.py's synthetic graph of that size (same seeds), built once per module run."""
    if size not in _CACHE:
        _CACHE.clear()  # one size at a time in host memory
        n, e = SIZES[size]
        ei = torch.randint(0, n, (2, e), generator=torch.Generator().manual_seed(1234567), dtype=torch.int64)
        x = torch.randn(n, 128, generator=torch.Generator().manual_seed(1234568))
        y = torch.randint(0, 128, (n,), generator=torch.Generator().manual_seed(1234569))
        _CACHE[size] = (ei, x, y)
    return _CACHE[size]


_CSR = {}


def csr_graph(size, kind, loops_mode=1):
    """oracle.large.CsrGraph (forward + transposed CSR with their weights) of the benchmark graph, kept across the tests of
    this module: its two stable sorts of 62 M edge keys are a third of the CPU time of an L-size case."""
    key = (size, kind, loops_mode)
    if key not in _CSR:
        for k in [k for k in _CSR if k[0] != size]:
            del _CSR[k]
        ei, x, _ = workload(size)
        _CSR[key] = OL.CsrGraph(ei, x.size(0), kind, loops_mode=loops_mode, threads=O.c_threads())
    return _CSR[key]


@pytest.mark.parametrize("size", ["S", "L"])
def test_fused_aggregate_transform_on_the_whole_benchmark_graph(dev, size):
    from rgb_experiment_amd import ops
    from rgb_experiment_amd.graph import clear_cache, get_graph
    ei, x, _ = workload(size)
    n = x.size(0)
    g = torch.Generator().manual_seed(7)
    W = torch.randn(128, 128, generator=g) / 128 ** 0.5
    Wr = torch.randn(128, 128, generator=g) / 128 ** 0.5
    b = torch.randn(128, generator=g)
    ei_d, x_d = ei.to(dev), x.to(dev)
    threads = O.c_threads()
    # GCN: A_hat x W^T + b
    rei, w = O.gcn_norm(ei, None, n)
    rowptr, col, perm = O.csr_from_edges(rei[1], rei[0], torch.arange(rei.size(1)), n)
    want = O.propagate_c_csr(rowptr, col, w[perm.long()].contiguous(), x, "add", threads) @ W.t() + b
    with torch.no_grad():
        got = ops.propagate_linear(x_d, get_graph(ei_d, n, 1), "gcn", W.to(dev), b.to(dev)).cpu()
    assert (got - want).abs().max().item() < TOL
    del rei, w, rowptr, col, perm, want, got
    # SAGEConv form: mean over the in-edges as given (no self-loops), root term in the same kernel
    rowptr, col, _ = O.csr_from_edges(ei[1], ei[0], torch.arange(ei.size(1)), n)
    want = O.propagate_c_csr(rowptr, col, None, x, "mean", threads) @ W.t() + b + x @ Wr.t()
    with torch.no_grad():
        got = ops.propagate_linear(x_d, get_graph(ei_d, n, 0), "mean", W.to(dev), b.to(dev), root_weight=Wr.to(dev)).cpu()
    assert (got - want).abs().max().item() < TOL
    clear_cache()


@pytest.mark.parametrize("size", ["S", "L"])
def test_appnp_k10_on_the_whole_benchmark_graph(dev, size):
    """BASELINE config 5 as stated: APPNP K = 10, alpha = 0.1, d = 128 over the WHOLE benchmark graph.
    rgbx_appnp_f32 (ops.appnp_propagate: ten launches with the teleport in the store) against ten iterations of the C
    restatement's propagate + teleport on the CPU (recurrence: reference models/pta.py:79-84, APPNP behind
    models/appnp_stack.py:29); then the whole APPNPStack model with K = 10 — eval-mode logits of ALL nodes against the
    same model evaluated on the CPU (dense layers in torch, the ten propagates in the C restatement). Tolerance 1e-4."""
    from rgb_experiment_amd import models as M
    from rgb_experiment_amd import ops
    from rgb_experiment_amd.graph import clear_cache, get_graph
    ei, x, y = workload(size)
    n = x.size(0)
    K, alpha = 10, 0.1
    cg = csr_graph(size, "gcn")  # gcn_norm + per-target CSR (oracle.large.CsrGraph, pinned by tests/test_oracle_large.py)

    def appnp_cpu(h):
        z = h
        for _ in range(K):
            z = (1 - alpha) * cg.forward(z) + alpha * h
        return z

    ei_d, x_d = ei.to(dev), x.to(dev)
    with torch.no_grad():
        got = ops.appnp_propagate(x_d, get_graph(ei_d, n, 1), K, alpha).cpu()
    want = appnp_cpu(x)
    assert (got - want).abs().max().item() < TOL
    del got, want
    if size == "L":  # the whole model at L: test_model_gradients_at_benchmark_size_L[appnpstack] (loss over all rows + every
        clear_cache()  # gradient against oracle/large.py); the eval-mode model on all rows is checked at S below
        return
    # the model of config 5: lin1 -> BatchNorm -> lin2 -> APPNP(K = 10), after two training steps
    torch.manual_seed(14530529)
    model = M.APPNPStack(input_dim=128, output_dim=128, hidden_unit=64, K=K, alpha=alpha, dropout_rate=0.5).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    y_d = y.to(dev)
    mask = (torch.arange(n, device=dev) % 5) < 3
    model.train()
    for _ in range(2):
        opt.zero_grad()
        ops.masked_ce_loss(model(x_d, ei_d)["emb"], y_d, mask).backward()
        opt.step()
    model.eval()
    with torch.no_grad():
        got = model(x_d, ei_d)["emb"].cpu()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    h = x @ sd["lin1.weight"].t() + sd["lin1.bias"]
    h = O.batch_norm(h, sd, "bn.", False)
    h = h @ sd["lin2.weight"].t() + sd["lin2.bias"]
    want = appnp_cpu(h)
    err = (got - want).abs().max().item()
    assert err < TOL, (size, err, want.abs().max().item())
    del model, opt
    clear_cache()
    torch.cuda.empty_cache()


MODEL_KW = {
    "gcn": dict(num_layers=2, hidden_unit=128, dropout_rate=0.5),
    "graphsage": dict(num_layers=2, hidden_unit=128, dropout_rate=0.5),
    "graphsage2": dict(num_layers=2, hidden_unit=128, dropout_rate=0.5),
    "gat": dict(num_layers=2, hidden_unit=16, heads=8, dropout_rate=0.5),
    "appnpstack": dict(hidden_unit=64, K=2, alpha=0.1, dropout_rate=0.5),  # K = 2: see oracle/sampled.py
    # SURVEY 8(f) callers of the same kernels
    "sgc": dict(K=2, cached=False),
    "gin": dict(num_layers=2, hidden_unit=128, dropout_rate=0.0),
    "dagnn": dict(hidden_dim=64, K=2, dropout_rate=0.0),
}


@pytest.mark.parametrize("name", ["gcn", "graphsage", "graphsage2", "gat", "appnpstack", "sgc", "gin", "dagnn"])
@pytest.mark.parametrize("size", ["S", "L"])
def test_model_logits_at_sampled_rows_of_the_benchmark_graph(dev, size, name):
    """BASELINE configs 2-5 in their one-GPU form: two training steps (so that weights, BatchNorm running statistics
    and biases are not at their initial values), then the eval-mode logits of sampled nodes against the oracle."""
    from rgb_experiment_amd import models as M
    from rgb_experiment_amd import ops
    from rgb_experiment_amd.graph import clear_cache
    ei, x, y = workload(size)
    n = x.size(0)
    cls = {"gcn": M.GCN, "graphsage": M.GraphSAGE, "graphsage2": M.GraphSAGE2, "gat": M.GAT,
           "appnpstack": M.APPNPStack, "sgc": M.SGC, "gin": M.GIN, "dagnn": M.DAGNN}[name]
    torch.manual_seed(14530529)
    model = cls(input_dim=128, output_dim=128, **MODEL_KW[name]).to(dev)
    ei_d, x_d, y_d = ei.to(dev), x.to(dev), y.to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    mask = (torch.arange(n, device=dev) % 5) < 3
    model.train()
    for _ in range(2):
        opt.zero_grad()
        ops.masked_ce_loss(model(x_d, ei_d)["emb"], y_d, mask).backward()
        opt.step()
    model.eval()
    n_targets = 96 if name in ("gcn", "appnpstack", "sgc", "dagnn") else 512
    targets = S.pick_targets(n, n_targets)
    with torch.no_grad():
        got = model(x_d, ei_d)["emb"][targets.to(dev)].cpu()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    kw = {k: v for k, v in MODEL_KW[name].items() if k not in ("hidden_unit", "hidden_dim", "dropout_rate", "cached")}
    want, info = S.sampled_logits(name, sd, x, ei, targets, **kw)
    err = (got - want).abs().max().item()
    assert err < TOL, (name, size, err, info)
    del model, opt
    clear_cache()
    torch.cuda.empty_cache()


# ---- backward at the BASELINE sizes -----------------------------------------------------------------------------

GRAD_KW = dict(MODEL_KW, appnpstack=dict(hidden_unit=64, K=10, alpha=0.1, dropout_rate=0.5))  # config 5 as stated
GRAD_TOL, GRAD_REL = 1e-4, 2e-3


def _hip_training_step(dev, name, ei, x, y, mask, route):
    """One train-mode forward + backward of the product model on the GPU. route 'kernel_loss': the loss inside the last
    conv's kernel where the model has that form (what experiment() and bench.py run); 'logits': log-probabilities
    materialised, NLLLoss on top (what a foreign loop runs). Returns (loss, {name: grad on the CPU}, initial state)."""
    from rgb_experiment_amd import models as M
    from rgb_experiment_amd.models._stack import masked_ce
    cls = {"gcn": M.GCN, "graphsage": M.GraphSAGE, "graphsage2": M.GraphSAGE2, "gat": M.GAT,
           "appnpstack": M.APPNPStack}[name]
    torch.manual_seed(14530529)
    model = cls(input_dim=128, output_dim=128, **GRAD_KW[name])
    with torch.no_grad():  # biases and BatchNorm affine parameters off their zero / one initial values
        g = torch.Generator().manual_seed(5)
        for k, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.to(dev).train()
    x_d, ei_d, y_d, m_d = x.to(dev), ei.to(dev), y.to(dev), mask.to(dev)
    if route == "kernel_loss":
        loss = masked_ce(model, {"x": x_d, "edge_index": ei_d}, y_d, m_d)[0]
    else:
        loss = torch.nn.functional.nll_loss(model(x_d, ei_d)["out"][m_d], y_d[m_d])
    loss.backward()
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().cpu() for k, p in model.named_parameters()}
    return float(loss.item()), grads, sd


def _check(rep, loss, ref_loss, what):
    print(f"gradient parity {what}: loss {loss:.7f} vs {ref_loss:.7f}; max |dg| {rep['max_abs']:.3e}, vs bound "
          f"{rep['max_vs_bound']:.3e}, relative {rep['max_rel']:.3e} ({rep['worst']})")
    assert abs(loss - ref_loss) < 1e-5 * max(1.0, abs(ref_loss)), (what, loss, ref_loss)
    assert rep["max_vs_bound"] < GRAD_TOL, (what, rep)
    assert rep["max_rel"] < GRAD_REL, (what, rep)


# both loss routes for the conv stacks, whose last layer can take the loss into its kernel; GAT (8 heads in, one head out:
# no epilogue at 128 classes through this route) and APPNP reach the same kernels on either route: one of them each
_GRAD_CASES_S = [(n, r) for n in ("gcn", "graphsage", "graphsage2") for r in ("kernel_loss", "logits")] + [
    ("gat", "kernel_loss"), ("appnpstack", "logits")]


@pytest.mark.parametrize("name,route", _GRAD_CASES_S)
def test_model_gradients_at_benchmark_size_S(dev, name, route):
    """S (|V| = 200 k, |E| = 4 M, d = 128): every parameter gradient against the FULL oracle (oracle.ref_cpu, the PyG
    dataflow under torch autograd: edge-sized temporaries of 2 GB each)."""
    from rgb_experiment_amd.graph import clear_cache
    ei, x, y = workload("S")
    n = x.size(0)
    mask = (torch.arange(n) % 5) < 3
    loss, grads, sd = _hip_training_step(dev, name, ei, x, y, mask, route)
    kw = GRAD_KW[name]
    fwd = {"gcn": lambda p: O.gcn_forward(p, x, ei, 2, True), "graphsage": lambda p: O.graphsage_forward(p, x, ei, 2, True),
           "graphsage2": lambda p: O.graphsage2_forward(p, x, ei, 2, True),
           "gat": lambda p: O.gat_forward(p, x, ei, 2, kw.get("heads", 8), True),
           "appnpstack": lambda p: O.appnp_stack_forward(p, x, ei, kw.get("K"), kw.get("alpha"), True)}[name]
    ref_sd = {k: v.clone().requires_grad_(v.is_floating_point() and "running_" not in k) for k, v in sd.items()}
    ref_loss = OL.masked_nll(fwd(ref_sd), y, mask)
    ref_loss.backward()
    # state_dict lists GATConv's lin_dst.weight next to lin_src.weight (one shared tensor: one parameter, one gradient)
    rep = OL.compare_grads(grads, {k: ref_sd[k].grad for k in grads})
    _check(rep, loss, ref_loss.item(), f"S {name} {route}")
    clear_cache()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("name", ["gcn", "graphsage", "graphsage2", "appnpstack"])
def test_model_gradients_at_benchmark_size_L(dev, name):
    """L (|V| = 2 M, |E| = 60 M, d = 128; BASELINE configs 4 and 5 and the headline GCN): every parameter gradient
    against oracle/large.py — propagate = C restatement over the CSR, its backward = the same over the transposed CSR,
    dense layers / BatchNorm / loss under CPU autograd."""
    from rgb_experiment_amd.graph import clear_cache
    ei, x, y = workload("L")
    n = x.size(0)
    mask = (torch.arange(n) % 5) < 3
    loss, grads, sd = _hip_training_step(dev, name, ei, x, y, mask, "kernel_loss")
    torch.cuda.empty_cache()
    graph = csr_graph("L", *{"gcn": ("gcn", 1), "appnpstack": ("gcn", 1), "graphsage": ("mean", 2),
                             "graphsage2": ("mean", 0)}[name])
    kw = {k: v for k, v in GRAD_KW[name].items() if k in ("num_layers", "K", "alpha")}
    ref_loss, ref_grads, _ = OL.loss_and_grads(name, sd, x, y, mask, graph, **kw)
    rep = OL.compare_grads(grads, ref_grads)
    _check(rep, loss, ref_loss, f"L {name}")
    clear_cache()
    torch.cuda.empty_cache()
