"""oracle/sampled.py: the oracle forward on the L-hop in-neighbourhood of sampled targets equals the oracle forward
on the whole graph at those targets — the property the full-size logit checks (tests/test_gpu_parity.py,
bench.py `parity.sampled_logits`) rest on. CPU only."""
import pytest
import torch

from oracle import ref_cpu as O
from oracle import sampled as S


def _graph(n=600, e=4200, f=12, seed=3):
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, n, (2, e), generator=g)
    loops = torch.randint(0, n, (15,), generator=g)
    ei = torch.cat([ei, torch.stack([loops, loops]), ei[:, :11]], dim=1)  # self-loops and duplicate edges
    ei[1, ei[1] == 7] = 8  # node 7: no in-edges at all
    return ei, torch.randn(n, f, generator=g)


def _state(name, f, c):
    from rgb_experiment_amd import models as M
    torch.manual_seed(5)
    if name == "gat":
        m = M.GAT(num_layers=2, hidden_unit=4, heads=3, input_dim=f, output_dim=c, dropout_rate=0.5)
    elif name == "appnpstack":
        m = M.APPNPStack(hidden_unit=16, input_dim=f, output_dim=c, K=2, alpha=0.1, dropout_rate=0.5)
    elif name == "sgc":
        m = M.SGC(input_dim=f, output_dim=c, K=2, cached=False)
    elif name == "gin":
        m = M.GIN(input_dim=f, output_dim=c, hidden_unit=16, num_layers=2, dropout_rate=0.5)
    elif name == "dagnn":
        m = M.DAGNN(input_dim=f, hidden_dim=16, output_dim=c, K=2, dropout_rate=0.5)
    else:
        cls = {"gcn": M.GCN, "graphsage": M.GraphSAGE, "graphsage2": M.GraphSAGE2}[name]
        m = cls(num_layers=2, hidden_unit=16, input_dim=f, output_dim=c, dropout_rate=0.5)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    for k in sd:  # running statistics that are not the identity
        if k.endswith("running_mean"):
            sd[k] = torch.randn_like(sd[k]) * 0.3
        if k.endswith("running_var"):
            sd[k] = torch.rand_like(sd[k]) + 0.5
    return sd


@pytest.mark.parametrize("name", ["gcn", "graphsage", "graphsage2", "gat", "appnpstack", "sgc", "gin", "dagnn"])
def test_sampled_subgraph_forward_equals_the_full_forward(name):
    ei, x = _graph()
    n, c = x.size(0), 5
    sd = _state(name, x.size(1), c)
    kw = {"gcn": dict(num_layers=2), "graphsage": dict(num_layers=2), "graphsage2": dict(num_layers=2),
          "gat": dict(num_layers=2, heads=3), "appnpstack": dict(K=2, alpha=0.1), "sgc": dict(K=2),
          "gin": dict(num_layers=2), "dagnn": dict(K=2)}[name]
    with torch.no_grad():
        full = S._forward(name, sd, x, ei, kw)["emb"]
    targets = torch.cat([S.pick_targets(n, 40), torch.tensor([7, 8])]).unique()
    got, info = S.sampled_logits(name, sd, x, ei, targets, **kw)
    assert info["sub_nodes"] < n + 2 and info["targets"] == targets.numel()
    assert (got - full[targets]).abs().max().item() < 2e-6, name


def test_in_degree_completion_is_what_makes_gcn_norm_exact():
    """Without the dummy in-edges the outermost sources have the wrong deg^-1/2 and the GCN rows differ."""
    ei, x = _graph()
    sd = _state("gcn", x.size(1), 5)
    targets = S.pick_targets(x.size(0), 30)
    with torch.no_grad():
        full = O.gcn_forward(sd, x, ei, 2)["emb"][targets]
        sub, ids, tpos, n_tot = S.khop_in_subgraph(ei, x.size(0), targets, 2, complete_in_degree=False)
        cut = O.gcn_forward(sd, S._sub_features(x, ids, n_tot), sub, 2)["emb"][tpos]
    assert (cut - full).abs().max().item() > 1e-3
