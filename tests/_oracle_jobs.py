"""The CPU-oracle halves of the slowest `-m gpu` cases, as named JOBS — test infrastructure only.

A BASELINE-size parity case has two halves that do not depend on each other: the HIP path on the GPU (milliseconds to
seconds) and the oracle on the host cores (seconds to tens of seconds: 62 M-edge stable sorts, OpenMP propagates, 300 Adam
steps under CPU autograd). Run one after the other they were 130 of the suite's 600 s. Both halves start from the same
SEEDED inputs (bench.py's synthetic workloads, torch.manual_seed'ed model states), so the oracle half needs nothing from the
GPU half: `tests/conftest.py` starts `python tests/_oracle_jobs.py <dir> <group>` processes at session start (CPU only: the
GPU is hidden from them), which work through the job list while the GPU-bound modules run, and a test asks `get(name)`:
the finished result if it is there, otherwise — the background run was not started, died, or is behind — THE SAME FUNCTION
inline. The numbers a test compares are identical either way; only the wall clock differs.

Nothing here is product code; nothing in the product imports it (tests/test_abi.py checks)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import large as OL  # noqa: E402
from oracle import ref_cpu as O  # noqa: E402

SIZES = {"S": (200_000, 4_000_000), "L": (2_000_000, 60_000_000)}
_WORK = {}
_CSR = {}


def workload(size):
    """bench.py's synthetic graph of that size (same seeds), one size at a time in host memory."""
    if size not in _WORK:
        _WORK.clear()
        n, e = SIZES[size]
        ei = torch.randint(0, n, (2, e), generator=torch.Generator().manual_seed(1234567), dtype=torch.int64)
        x = torch.randn(n, 128, generator=torch.Generator().manual_seed(1234568))
        y = torch.randint(0, 128, (n,), generator=torch.Generator().manual_seed(1234569))
        _WORK[size] = (ei, x, y)
    return _WORK[size]


def csr_graph(size, kind, loops_mode=1):
    """oracle.large.CsrGraph of the benchmark graph, kept while the size stays (two stable sorts of 62 M keys each)."""
    key = (size, kind, loops_mode)
    if key not in _CSR:
        for k in [k for k in _CSR if k[0] != size]:
            del _CSR[k]
        ei, x, _ = workload(size)
        if kind == "gat":  # the same rewritten edge list as my_SAGEConv's: share the CSRs
            _CSR[key] = csr_graph(size, "mean", 2).as_gat()
        else:
            _CSR[key] = OL.CsrGraph(ei, x.size(0), kind, loops_mode=loops_mode, threads=O.c_threads())
    return _CSR[key]


MODEL_KW = {
    "gcn": dict(num_layers=2, hidden_unit=128, dropout_rate=0.5),
    "graphsage": dict(num_layers=2, hidden_unit=128, dropout_rate=0.5),
    "graphsage2": dict(num_layers=2, hidden_unit=128, dropout_rate=0.5),
    "gat": dict(num_layers=2, hidden_unit=16, heads=8, dropout_rate=0.5),
    "appnpstack": dict(hidden_unit=64, K=10, alpha=0.1, dropout_rate=0.5),  # BASELINE config 5 as stated
}


MODEL_KW_S = MODEL_KW  # (a K = 4 APPNP at S was tried while the suite was over budget; the budget problem was the thread count)


def initial_state(name, size="L"):
    """The seeded initial state_dict of the gradient cases (biases and BatchNorm affine parameters off their zero / one
    initial values) — built on the CPU, the same bits wherever it is called."""
    from rgb_experiment_amd import models as M
    cls = {"gcn": M.GCN, "graphsage": M.GraphSAGE, "graphsage2": M.GraphSAGE2, "gat": M.GAT, "appnpstack": M.APPNPStack}[name]
    torch.manual_seed(14530529)
    model = cls(input_dim=128, output_dim=128, **(MODEL_KW_S if size == "S" else MODEL_KW)[name])
    with torch.no_grad():
        g = torch.Generator().manual_seed(5)
        for k, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    return model, {k: v.detach().clone() for k, v in model.state_dict().items()}


def train_mask(n):
    return (torch.arange(n) % 5) < 3


# ---- the jobs ------------------------------------------------------------------------------------------------------

def grads_S(name):
    """(loss, {parameter: gradient}) of one training step at S under the FULL oracle (PyG dataflow, torch autograd)."""
    ei, x, y = workload("S")
    mask = train_mask(x.size(0))
    _, sd = initial_state(name, "S")
    kw = MODEL_KW_S[name]
    fwd = {"gcn": lambda p: O.gcn_forward(p, x, ei, 2, True), "graphsage": lambda p: O.graphsage_forward(p, x, ei, 2, True),
           "graphsage2": lambda p: O.graphsage2_forward(p, x, ei, 2, True),
           "gat": lambda p: O.gat_forward(p, x, ei, 2, kw.get("heads", 8), True),
           "appnpstack": lambda p: O.appnp_stack_forward(p, x, ei, kw.get("K"), kw.get("alpha"), True)}[name]
    ref_sd = {k: v.clone().requires_grad_(v.is_floating_point() and "running_" not in k) for k, v in sd.items()}
    loss = OL.masked_nll(fwd(ref_sd), y, mask)
    loss.backward()
    return loss.item(), {k: v.grad for k, v in ref_sd.items() if v.requires_grad and v.grad is not None}


_L_GRAPH = {"gcn": ("gcn", 1), "appnpstack": ("gcn", 1), "graphsage": ("mean", 2), "graphsage2": ("mean", 0), "gat": ("gat", 2)}


def grads_L(name):
    """The same at L under oracle/large.py (propagate and its adjoint through the C restatement over the CSRs)."""
    ei, x, y = workload("L")
    mask = train_mask(x.size(0))
    _, sd = initial_state(name)
    graph = csr_graph("L", *_L_GRAPH[name])
    kw = {k: v for k, v in MODEL_KW[name].items() if k in ("num_layers", "K", "alpha", "heads")}
    loss, grads, _ = OL.loss_and_grads(name, sd, x, y, mask, graph, **kw)
    return loss, grads


def appnp_k10(size):
    """APPNP K = 10, alpha = 0.1 of the workload's features over the whole graph: ten iterations of the C restatement."""
    _, x, _ = workload(size)
    cg = csr_graph(size, "gcn")
    z = x
    for _ in range(10):
        z = 0.9 * cg.forward(z) + 0.1 * x
    return z


def fused_expect(size, form):
    """form 'gcn': A_hat x W^T + b; form 'sage': mean_j(x_j) W^T + b + x Wr^T (edges as given) — of the whole benchmark graph
    with the test's seeded W, Wr, b."""
    _, x, _ = workload(size)
    g = torch.Generator().manual_seed(7)
    W = torch.randn(128, 128, generator=g) / 128 ** 0.5
    Wr = torch.randn(128, 128, generator=g) / 128 ** 0.5
    b = torch.randn(128, generator=g)
    if form == "gcn":
        return csr_graph(size, "gcn").forward(x) @ W.t() + b
    return csr_graph(size, "mean", 0).forward(x) @ W.t() + b + x @ Wr.t()


def cora300():
    """Training losses of 300 Adam steps of the Cora-shaped GCN (BASELINE config 1) under the oracle's autograd."""
    from rgb_experiment_amd.models import GCN
    from rgb_experiment_amd.utils import get_whole_mask
    n, pairs, f, c = 2708, 5278, 1433, 7
    gen = torch.Generator().manual_seed(1234567)
    a = torch.randint(0, n, (pairs,), generator=gen)
    b = (a + 1 + torch.randint(0, n - 1, (pairs,), generator=gen)) % n
    ei = torch.cat([torch.stack([a, b]), torch.stack([b, a])], dim=1)
    x = torch.zeros(n, f)
    x.scatter_(1, torch.randint(0, f, (n, 18), generator=gen), 1.0)
    y = torch.randint(0, c, (n,), generator=gen)
    xn = x / x.sum(1, keepdim=True).clamp(min=1)
    mask = get_whole_mask(y, "6-2-2", 123456789)[0]
    torch.manual_seed(14530529)
    ref = GCN(num_layers=2, hidden_unit=64, input_dim=f, output_dim=c, dropout_rate=0.5)
    params = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in ref.state_dict().items()}
    opt = torch.optim.Adam([params[k] for k, _ in ref.named_parameters()], lr=0.01)
    losses = []
    for _ in range(300):
        opt.zero_grad()
        out = O.gcn_forward(params, xn, ei, 2, training=True)["out"]
        loss = torch.nn.functional.nll_loss(out[mask], y[mask])
        loss.backward()
        opt.step()
        losses.append(loss.item())
    return losses


# job name -> (function, arguments); GROUPS: what one background process works through, in the order the suite asks
JOBS = {"cora300": (cora300, ())}
for _n in ("gcn", "graphsage", "graphsage2", "gat", "appnpstack"):
    JOBS[f"grads_S_{_n}"] = (grads_S, (_n,))
    JOBS[f"grads_L_{_n}"] = (grads_L, (_n,))
for _s in ("S", "L"):
    JOBS[f"appnp_k10_{_s}"] = (appnp_k10, (_s,))
    JOBS[f"fused_expect_gcn_{_s}"] = (fused_expect, (_s, "gcn"))
    JOBS[f"fused_expect_sage_{_s}"] = (fused_expect, (_s, "sage"))
# one background process per group; a group's jobs share what they build (workload, CSRs): the gcn_norm graph in one, the two
# mean graphs (edges as given; remove + add self-loops, which GAT shares) in another
GROUPS = {
    "cora": ["cora300"],
    "small": ["fused_expect_gcn_S", "fused_expect_sage_S", "appnp_k10_S"] + [
        f"grads_S_{n}" for n in ("gcn", "graphsage", "graphsage2", "gat", "appnpstack")],
    "large_gcn": ["fused_expect_gcn_L", "appnp_k10_L", "grads_L_gcn", "grads_L_appnpstack"],
    "large_mean": ["fused_expect_sage_L", "grads_L_graphsage2", "grads_L_graphsage", "grads_L_gat"],
}


def compute(name):
    fn, args = JOBS[name]
    return fn(*args)


def _path(d, name):
    return os.path.join(d, name + ".pt")


def get(name, wait_s=600.0):
    """The job's result: from the background run when RGBX_ORACLE_BG names its directory and the job is (or, while a
    background process is still alive, becomes) available there; otherwise computed here and now."""
    d = os.environ.get("RGBX_ORACLE_BG")
    if d and os.path.isdir(d):
        t_end = time.time() + wait_s
        while time.time() < t_end:
            if os.path.exists(_path(d, name)):
                return torch.load(_path(d, name))
            if os.path.exists(_path(d, name) + ".failed") or not _alive(d, name):
                break
            time.sleep(0.2)
        if os.path.exists(_path(d, name)):
            return torch.load(_path(d, name))
    return compute(name)


def _alive(d, name):
    """The background process whose group holds job `name` is still running (conftest writes <group>.pid when it starts it;
    a zombie counts as gone)."""
    for group, names in GROUPS.items():
        if name in names:
            try:
                pid = int(open(os.path.join(d, f"{group}.pid")).read())
                os.kill(pid, 0)
                with open(f"/proc/{pid}/stat") as f:
                    return f.read().rsplit(")", 1)[1].split()[0] != "Z"
            except (OSError, ValueError, IndexError):
                return False
    return False


def main(d, group):
    log = open(os.path.join(d, f"{group}.log"), "w")
    for name in GROUPS[group]:
        t0 = time.time()
        try:
            res = compute(name)
            torch.save(res, _path(d, name) + ".tmp")
            os.replace(_path(d, name) + ".tmp", _path(d, name))
            print(f"{name}: {time.time() - t0:.1f} s", file=log, flush=True)
        except Exception as exc:  # noqa: BLE001 - the test computes the job inline and shows the real error
            open(_path(d, name) + ".failed", "w").write(repr(exc))
            print(f"{name}: FAILED {exc!r}", file=log, flush=True)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
