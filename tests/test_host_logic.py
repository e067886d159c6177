"""Host-side pieces either side of the hot path: split masks (bit-exact vs the reference, golden G4),
subgraph relabelling (G5), metrics (G6), edge utilities, Data container, experiment() argument handling."""
import numpy as np
import pytest
import torch

import rgb_experiment_amd as R
from rgb_experiment_amd import utils as U
from rgb_experiment_amd.models import REGISTRY


@pytest.mark.parametrize("case", ["small", "cora_shaped"])
@pytest.mark.parametrize("ratio,seed", [("6-2-2", 123456789), ("5-2-3", 14530529), ("1-1-3", 1234567)])
def test_g4_masks_bit_exact(golden, case, ratio, seed):
    y = torch.from_numpy(golden[f"g4/{case}/y"])
    got = torch.stack(U.get_whole_mask(y, ratio, seed)).numpy()
    assert np.array_equal(got, golden[f"g4/{case}/whole/{ratio}/{seed}"])
    got = torch.stack(U.get_classification_mask(y, ratio, seed)).numpy()
    assert np.array_equal(got, golden[f"g4/{case}/classification/{ratio}/{seed}"])


@pytest.mark.parametrize("case", ["small", "cora_shaped"])
def test_g4_random_mask_train_part(golden, case):
    y = torch.from_numpy(golden[f"g4/{case}/y"])
    train, val, test = U.get_random_mask(y, 5, 10, 20, 1234567)
    assert np.array_equal(train.numpy(), golden[f"g4/{case}/random_train/5/1234567"])
    assert int(val.sum()) == 10 and int(test.sum()) == 20
    assert not (train & val).any() and not (train & test).any() and not (val & test).any()
    assert not (y[train | val | test] == -1).any()


def test_g5_node_induced_subgraph(golden):
    ei = torch.from_numpy(golden["g5/edge_index"])
    n = int(golden["g5/num_nodes"])
    picked = golden["g5/nodes_list"].tolist()
    assert np.array_equal(U.node_induced_subgraph(n, picked, ei, True).numpy(), golden["g5/list_reorder"])
    assert np.array_equal(U.node_induced_subgraph(n, picked, ei, False).numpy(), golden["g5/list_keep"])
    m = torch.from_numpy(golden["g5/nodes_mask"])
    assert np.array_equal(U.node_induced_subgraph(n, m, ei, True).numpy(), golden["g5/mask_reorder"])


def test_g6_compare_pred_label(golden):
    pred, label = torch.from_numpy(golden["g6/pred"]), torch.from_numpy(golden["g6/label"])
    res = R.compare_pred_label(pred, label, True)
    got = [res[k] for k in ("ACC", "precision_score", "recall_score", "f1_macro", "f1_micro")]
    assert np.allclose(got, golden["g6/metrics"], atol=1e-12)
    res0 = R.compare_pred_label(pred, label, False)  # reference raises UnboundLocalError here
    assert res0["ACC"] == res["ACC"] and res0["f1_macro"] == 0 and res0["f1_micro"] == 0


def test_metrics_without_sklearn_agree(golden, monkeypatch):
    import builtins
    from rgb_experiment_amd import itexperiments as it
    pred, label = golden["g6/pred"], golden["g6/label"]
    with_sk = it._macro_prf(label, pred)
    real_import = builtins.__import__

    def no_sklearn(name, *a, **k):
        if name.startswith("sklearn"):
            raise ImportError(name)
        return real_import(name, *a, **k)

    monkeypatch.setattr(builtins, "__import__", no_sklearn)
    assert np.allclose(it._macro_prf(label, pred), with_sk, atol=1e-12)


def test_edge_utils():
    ei = torch.tensor([[2, 0, 0, 1, 1], [0, 1, 1, 1, 2]])
    und = U.to_undirected(ei, 3)
    assert und.tolist() == [[0, 0, 1, 1, 1, 2, 2], [1, 2, 0, 1, 2, 0, 1]]
    assert U.coalesce(ei, 3).tolist() == [[0, 1, 1, 2], [1, 1, 2, 0]]
    assert U.remove_self_loops(ei).tolist() == [[2, 0, 0, 1], [0, 1, 1, 2]]
    assert U.add_remaining_self_loops(ei, 3).tolist() == [[2, 0, 0, 1, 0, 1, 2], [0, 1, 1, 2, 0, 1, 2]]
    assert U.to_undirected(torch.zeros(2, 0, dtype=torch.long), 3).shape == (2, 0)


def test_data_container():
    d = R.Data(x=torch.randn(4, 3), y=torch.tensor([0, 1, 0, 1]), edge_index=torch.tensor([[0], [1]]))
    assert d.num_nodes == 4 and d.num_node_features == 3 and d.num_edges == 1
    c = d.clone()
    c.x[0, 0] = 99.0
    assert d.x[0, 0] != 99.0
    d.train_mask = torch.tensor([True, False, False, False])
    assert d.to("cpu").train_mask.tolist() == [True, False, False, False]
    assert "edge_index" in repr(d)


def test_registry_and_constructor_signatures():
    assert sorted(REGISTRY) == ["appnpstack", "dagnn", "gat", "gcn", "gin", "graphsage", "graphsage2", "mlp", "pta",
                                "sgc"]
    m = REGISTRY["gcn"](input_dim=5, output_dim=3, **R.InitialParameters.defaults_for("GCN"))
    assert [k for k in m.state_dict() if k.startswith("convs.0")] == ["convs.0.bias", "convs.0.lin.weight"]
    assert m.state_dict()["convs.0.lin.weight"].shape == (64, 5)
    g = REGISTRY["gat"](input_dim=5, output_dim=3, **R.InitialParameters.defaults_for("gat"))
    sd = g.state_dict()
    assert sd["convs.0.att_src"].shape == (1, 8, 8) and sd["convs.0.lin_src.weight"].shape == (64, 5)
    assert sd["convs.1.att_dst"].shape == (1, 1, 3) and sd["convs.1.bias"].shape == (3,)
    assert sd["bns.0.weight"].shape == (64,)
    s = REGISTRY["graphsage"](input_dim=5, output_dim=3, **R.InitialParameters.defaults_for("graphsage"))
    assert {"convs.0.lin_l.bias", "convs.0.lin_r.bias"} <= set(s.state_dict())
    s2 = REGISTRY["graphsage2"](input_dim=5, output_dim=3, **R.InitialParameters.defaults_for("graphsage2"))
    assert "convs.0.lin_l.bias" in s2.state_dict() and "convs.0.lin_r.bias" not in s2.state_dict()
    a = REGISTRY["appnpstack"](input_dim=5, output_dim=3, **R.InitialParameters.defaults_for("appnpstack"))
    assert a.conv.K == 10 and a.conv.alpha == 0.1 and "lin1.weight" in a.state_dict()


def _toy(n=60, f=6, c=3, seed=0):
    g = torch.Generator().manual_seed(seed)
    return R.Data(x=torch.randn(n, f, generator=g), y=torch.randint(0, c, (n,), generator=g),
                  edge_index=torch.randint(0, n, (2, 200), generator=g))


def test_experiment_mlp_runs_on_cpu_and_returns_metric_dict():
    """MLP has no message passing, so the harness itself can be exercised without a GPU."""
    res = R.experiment({"num_layers": 2, "hidden_unit": 8, "dropout_rate": 0.5}, specify_data=True, data=_toy(),
                       model_name="MLP", epoch=5, learning_rate=0.01, use_cpu=True, print_print=False,
                       need_to_reappear=True, return_model=True)
    assert set(res) >= {"ACC", "precision_score", "recall_score", "f1_macro", "f1_micro"}
    assert 0.0 <= res["ACC"] <= 1.0 and len(res["history"]["val_acc"]) == 5
    out = res["model"](_toy().x)
    assert set(out) == {"out", "emb", "x"} and torch.allclose(out["out"].exp().sum(1), torch.ones(60), atol=1e-5)


def test_experiment_early_stopping_and_seeding_are_deterministic():
    kw = dict(specify_data=True, data=_toy(), model_name="mlp", epoch=60, learning_rate=0.05, use_cpu=True,
              print_print=False, need_to_reappear=True, early_stopping=2, begin_early_stopping=3,
              return_model=True, need_all_metrics=False)
    a = R.experiment({"num_layers": 2, "hidden_unit": 8, "dropout_rate": 0.5}, **kw)
    b = R.experiment({"num_layers": 2, "hidden_unit": 8, "dropout_rate": 0.5}, **kw)
    assert a["ACC"] == b["ACC"] and a["history"]["train_loss"] == b["history"]["train_loss"]
    assert len(a["history"]["train_loss"]) < 60  # stopped early


def test_experiment_rejects_cpu_for_graph_models_and_out_of_scope_names():
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        R.experiment(R.InitialParameters.defaults_for("gcn"), specify_data=True, data=_toy(), model_name="GCN",
                     use_cpu=True, print_print=False)
    with pytest.raises(NotImplementedError):
        R.experiment({}, specify_data=True, data=_toy(), model_name="FAGCN", print_print=False)
    with pytest.raises(ValueError):
        R.experiment({}, specify_data=True, data=_toy(), model_name="nope", print_print=False)


def test_experiment_mask_remake_rules():
    d = _toy()
    m = torch.zeros(60, dtype=torch.bool)
    d.train_mask, d.val_mask, d.test_mask = m.clone(), m.clone(), m.clone()
    d.train_mask[:30], d.val_mask[30:45], d.test_mask[45:] = True, True, True
    from rgb_experiment_amd.itexperiments import _masks_usable, _must_remake_masks
    assert _masks_usable(d)
    # default = the reference (:210, a condition that is always true): supplied masks are remade, valid or not
    assert _must_remake_masks(d, False, False) and _must_remake_masks(d, True, False)
    # the opt-in fix keeps valid masks; remake_data_mask still wins
    assert not _must_remake_masks(d, False, True) and _must_remake_masks(d, True, True)
    d.val_mask = None
    assert not _masks_usable(d) and _must_remake_masks(d, False, True)


def test_experiment_remakes_supplied_masks_by_default():
    """Same inputs, same results as the reference: with specify_data=True the masks the run uses are
    get_whole_mask(y, ratio, seed) (reference :210-215), whatever the Data object carried."""
    from rgb_experiment_amd.utils import get_whole_mask
    d = _toy()
    m = torch.zeros(60, dtype=torch.bool)
    d.train_mask, d.val_mask, d.test_mask = m.clone(), m.clone(), m.clone()
    d.train_mask[:2], d.val_mask[2:4], d.test_mask[4:6] = True, True, True  # 2 rows each: would show in the result
    kw = dict(specify_data=True, model_name="MLP", epoch=3, learning_rate=0.01, use_cpu=True, print_print=False,
              need_to_reappear=True, return_model=True)
    res = R.experiment({"num_layers": 2, "hidden_unit": 8, "dropout_rate": 0.5}, data=d, **kw)
    want = get_whole_mask(d.y, "6-2-2", 123456789)
    assert int(want[2].sum()) > 2
    fresh = _toy()
    fresh.train_mask, fresh.val_mask, fresh.test_mask = want
    kept = R.experiment({"num_layers": 2, "hidden_unit": 8, "dropout_rate": 0.5}, data=fresh,
                        keep_valid_data_mask=True, **kw)
    assert res["history"]["train_loss"] == kept["history"]["train_loss"] and res["ACC"] == kept["ACC"]
    own = R.experiment({"num_layers": 2, "hidden_unit": 8, "dropout_rate": 0.5}, data=d, keep_valid_data_mask=True, **kw)
    assert own["history"]["train_loss"] != res["history"]["train_loss"]  # the 2-row masks were honoured here


def test_rd2pd_roundtrip(tmp_path):
    d = _toy()
    folder = tmp_path / "toy"
    folder.mkdir()
    np.save(folder / "x.npy", d.x.numpy())
    np.save(folder / "y.npy", d.y.numpy())
    np.save(folder / "edge_index.npy", d.edge_index.numpy())
    ds = R.RD2PD("toy", str(tmp_path), split_ratio="6-2-2", split_seed=7, remove_self_loop=True,
                 remove_duplicate_edges=True)
    want = U.get_whole_mask(d.y, "6-2-2", 7)
    assert torch.equal(ds.data.train_mask, want[0]) and ds.num_nodes == 60
    ei = ds.data.edge_index
    assert (ei[0] != ei[1]).all() and torch.unique(ei[0] * 60 + ei[1]).numel() == ei.size(1)


def test_feature_normalisation_modes():
    from rgb_experiment_amd.itexperiments import _normalize_features
    x = torch.tensor([[1., 3.], [2., 2.], [0., 0.]])
    assert torch.allclose(_normalize_features(x, "row", None), torch.tensor([[.25, .75], [.5, .5], [0., 0.]]))
    mm = _normalize_features(x, "col", "MinMaxScalar")
    assert torch.allclose(mm, torch.tensor([[.5, 1.], [1., 2 / 3], [0., 0.]]))
    st = _normalize_features(x, "col", "StandardScalar")
    assert torch.allclose(st.mean(0), torch.zeros(2), atol=1e-6)


def test_row_split_plan_covers_long_rows_exactly():
    """Hub-row chunk plan (graph.make_row_split): pure index arithmetic, checked on CPU tensors."""
    from rgb_experiment_amd.graph import make_row_split
    deg = torch.tensor([3, 0, 2500, 1024, 1025, 7, 4096, 1])
    rowptr = torch.zeros(len(deg) + 1, dtype=torch.int32)
    rowptr[1:] = torch.cumsum(deg, 0)
    assert make_row_split(rowptr, threshold=5000) is None
    sp = make_row_split(rowptr, threshold=1024)
    assert sp["long_row"].tolist() == [2, 4, 6] and sp["n_long"] == 3
    assert sp["long_chunk_ptr"].tolist() == [0, 3, 5, 9] and sp["n_chunks"] == 9
    assert sp["chunk_row"].tolist() == [2, 2, 2, 4, 4, 6, 6, 6, 6]
    for r, row in enumerate(sp["long_row"].tolist()):
        c0, c1 = sp["long_chunk_ptr"][r].item(), sp["long_chunk_ptr"][r + 1].item()
        b, e = sp["chunk_begin"][c0:c1].tolist(), sp["chunk_end"][c0:c1].tolist()
        assert b[0] == rowptr[row].item() and e[-1] == rowptr[row + 1].item()
        assert all(x == y for x, y in zip(e[:-1], b[1:])) and all(0 < y - x <= 1024 for x, y in zip(b, e))


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_g7_pta_forward_and_loss(golden, mode):
    """PTA.forward / loss_function (reference models/pta.py:41-77) reproduce the reference's outputs."""
    from rgb_experiment_amd.models import PTA
    model = PTA(nfeat=6, nhid=5, nclass=4, dropout=0.0, epsilon=100, K=3, alpha=0.1, mode=mode)
    state = {k.split("state/")[1]: torch.from_numpy(v) for k, v in golden.items() if k.startswith(f"g7/mode{mode}/state/")}
    model.load_state_dict(state)
    x, y_soft = torch.from_numpy(golden["g7/x"]), torch.from_numpy(golden["g7/y_soft"])
    model.train()
    y_hat = model(x)
    assert torch.allclose(y_hat, torch.from_numpy(golden[f"g7/mode{mode}/forward"]), atol=1e-6)
    for epoch in (0, 7, 150):
        want = float(golden[f"g7/mode{mode}/train_loss/{epoch}"])
        assert abs(model.loss_function(y_hat, y_soft, epoch).item() - want) < 1e-6 * max(1.0, abs(want))
    model.eval()
    want = float(golden[f"g7/mode{mode}/eval_loss"])
    assert abs(model.loss_function(model(x), y_soft).item() - want) < 1e-6 * max(1.0, abs(want))


def test_pta_seeded_initialisation_equals_the_reference(golden):
    """The reference allocates PTA's Linear parameters with torch.randn (models/pta.py:19,21), which advances the
    global RNG before kaiming_uniform_ draws: under the same seed the product must start from the same weights
    (golden G7 was generated after torch.manual_seed(77), modes 0, 1, 2 built one after the other)."""
    from rgb_experiment_amd.models import PTA
    torch.manual_seed(77)
    x = torch.from_numpy(golden["g7/x"])
    for mode in (0, 1, 2):
        model = PTA(nfeat=6, nhid=5, nclass=4, dropout=0.0, epsilon=100, K=3, alpha=0.1, mode=mode)
        for k, v in model.state_dict().items():
            assert torch.equal(v, torch.from_numpy(golden[f"g7/mode{mode}/state/{k}"])), (mode, k)
        model.train()
        model(x)  # the generator ran a training forward (dropout 0.0) between the constructions
        model.eval()
        model(x)


def test_experiment_refuses_masks_that_select_unlabelled_nodes():
    """A -1 label inside a mask: the reference's NLLLoss raises; the masked loss kernels would drop the row
    silently, so experiment() refuses up front."""
    d = _toy()
    d.y[3] = -1
    d.train_mask = torch.zeros(60, dtype=torch.bool)
    d.train_mask[:30] = True
    d.val_mask, d.test_mask = torch.zeros(60, dtype=torch.bool), torch.zeros(60, dtype=torch.bool)
    d.val_mask[30:45] = True
    d.test_mask[45:] = True
    with pytest.raises(RuntimeError, match="label is outside"):
        R.experiment({"num_layers": 2, "hidden_unit": 8, "dropout_rate": 0.5}, specify_data=True, data=d,
                     model_name="MLP", epoch=2, use_cpu=True, print_print=False, keep_valid_data_mask=True)


def test_experiment_accepts_index_list_masks():
    """Masks given as node-index lists (reference :193-209) are honoured (MLP: runs on the CPU)."""
    d = _toy()
    d.train_mask, d.val_mask, d.test_mask = list(range(0, 30)), list(range(30, 45)), list(range(45, 60))
    res = R.experiment({"num_layers": 2, "hidden_unit": 8, "dropout_rate": 0.5}, specify_data=True, data=d,
                       model_name="MLP", epoch=3, learning_rate=0.01, use_cpu=True, print_print=False,
                       keep_valid_data_mask=True)
    assert 0.0 <= res["ACC"] <= 1.0
    from rgb_experiment_amd.itexperiments import _as_bool_mask
    m = _as_bool_mask([1, 3], 5, "cpu")
    assert m.tolist() == [False, True, False, True, False]
    assert _as_bool_mask(torch.tensor([True, False]), 4, "cpu").tolist() == [True, False, False, False]


def test_model_output_is_the_reference_dict_with_lazy_log_probs():
    """models/_stack.ModelOutput: {'out', 'emb', 'x'} as the reference's forward returns (models/gcn.py:31), with
    'out' / 'x' = log_softmax(emb) produced on first access."""
    import torch
    from rgb_experiment_amd.models._stack import model_output
    z = torch.randn(6, 4)
    res = model_output(z)
    assert isinstance(res, dict) and res["emb"] is z
    assert "out" in res and "x" in res and "emb" in res and "nope" not in res
    assert list(res.keys()) == ["out", "emb", "x"] and len(res) == 3
    assert not dict.__contains__(res, "out")          # nothing computed yet
    want = torch.log_softmax(z, dim=1)
    assert torch.equal(res["out"], want) and res["x"] is res["out"]
    assert dict.__contains__(res, "out")              # cached
    assert torch.equal(model_output(z).get("x"), want) and model_output(z).get("nope", 5) == 5
    assert {k: v.shape for k, v in model_output(z).items()} == {k: z.shape for k in ("out", "emb", "x")}
    assert torch.equal(dict(model_output(z))["out"], want)
    try:
        res["nope"]
    except KeyError:
        pass
    else:
        raise AssertionError("missing key must raise KeyError")
    fresh = model_output(z)
    assert "'out'" in repr(fresh) and set(fresh.copy()) == {"out", "emb", "x"}
    assert torch.equal(model_output(z).pop("out"), want)


def test_native_shuffle_is_pythons_random_shuffle_bit_for_bit():
    """rgbx_py_random_shuffle_i64 (csrc/pyshuffle.hip, host code) against `random.seed(seed); random.shuffle(list(range(n)))`:
    the shuffle behind the reference's split masks (utils/mask.py:66-102) — seeds below and above 2^32 (one and two key
    words), negative, the int64 extremes; lengths around powers of two (the rejection loop of randbelow) and 0 / 1."""
    import random
    from rgb_experiment_amd.utils import mask as M
    for seed in (0, 1, 123456789, 14530529, 2**32 - 1, 2**32, 2**40 + 7, -5, -2**63, 2**63 - 1, 1234567):
        for n in (0, 1, 2, 3, 7, 100, 1000, 4097, 65536, 65537, 100003):
            random.seed(seed)
            ref = list(range(n))
            random.shuffle(ref)
            assert M._py_shuffled_range(n, seed).tolist() == ref, (seed, n)
    # seeds the C++ path does not take (beyond int64, non-int) fall back to `random` itself
    random.seed(2**70 + 1)
    ref = list(range(50))
    random.shuffle(ref)
    assert M._py_shuffled_range(50, 2**70 + 1).tolist() == ref
    # and the module-level generator is left seeded as the reference leaves it
    M._py_shuffled_range(10, 42)
    a = random.random()
    random.seed(42)
    assert a == random.random()


def test_rccl_environment_for_ranks_that_share_a_device(monkeypatch):
    """dist/sharing.py: ranks share a device when two of them published the same identity (not when counts say so); then
    every rank names its own NCCL_HOSTID (RCCL's duplicate-GPU check compares host identities) and the loopback interface;
    values the caller exported stay."""
    import os

    from rgb_experiment_amd.dist import sharing
    assert not sharing.share_a_device(["h|0:5:0", "h|0:15:0", "h|0:65:0"])
    assert sharing.share_a_device(["h|0:5:0", "h|0:15:0", "h|0:5:0"])
    assert not sharing.share_a_device(["h|0:5:0"])
    for k in sharing.rccl_env(0):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("NCCL_SOCKET_IFNAME", "eth7")
    sharing.apply_env(1)
    assert os.environ["NCCL_HOSTID"] == sharing.rccl_env(1)["NCCL_HOSTID"] != sharing.rccl_env(0)["NCCL_HOSTID"]
    assert os.environ["NCCL_SOCKET_IFNAME"] == "eth7" and os.environ["NCCL_IB_DISABLE"] == "1"
    for k in sharing.rccl_env(0):
        monkeypatch.delenv(k, raising=False)


def test_whole_mask_with_more_classes_than_training_rows_raises_instead_of_spinning():
    """Reference utils/mask.py:16-21 bumps the seed until the train part contains every class — for ever when it has fewer
    rows than there are classes (found by a random sweep of experiment(): 33 nodes, 130 classes). Same inputs: a ValueError."""
    import pytest
    import torch

    from rgb_experiment_amd.utils import mask as M
    y = torch.arange(40) % 30  # 30 classes on 40 nodes
    with pytest.raises(ValueError, match="never end"):
        M.get_whole_mask(y, "6-2-2", seed=1)
    y2 = torch.arange(400) % 3  # feasible: returns after a few seeds at most
    tr, va, te = M.get_whole_mask(y2, "6-2-2", seed=1)
    assert M.check_train_containing(tr, y2) and int(tr.sum() + va.sum() + te.sum()) == 400


def test_whole_mask_seed_cap_is_a_keyword():
    """max_seed_tries: the cap on the reference's unbounded loop over seeds is the caller's to set; None
    restores the reference's behaviour (not exercised to the end here: it would not return)."""
    import pytest
    import torch

    from rgb_experiment_amd.utils import mask as M
    y = torch.cat([torch.zeros(300, dtype=torch.int64), torch.tensor([1, 1])])  # class 1: two nodes, 60 % train part
    fails = lambda s: not M.check_train_containing(M.get_order("6-2-2", torch.arange(302), 302, s)[0], y)
    first = next(s for s in range(1, 2000) if fails(s) and fails(s + 1) and fails(s + 2))
    with pytest.raises(ValueError, match="2 seeds tried"):
        M.get_whole_mask(y, "6-2-2", seed=first, max_seed_tries=2)  # seed, seed + 1, seed + 2 all fail
    tr, _, _ = M.get_whole_mask(y, "6-2-2", seed=first, max_seed_tries=None)  # the reference's loop: ends at the first good seed
    assert M.check_train_containing(tr, y)


def test_the_shuffle_leaves_the_module_generator_seeded_but_not_advanced():
    """The reference's `random.seed(seed); random.shuffle(...)` leaves the module-level generator seeded AND advanced by the
    shuffle's draws; the C++ restatement leaves it seeded only (documented in utils/mask._py_shuffled_range: nothing on the
    path draws from it before the next reseed). Pinned here so that a caller who starts to depend on the difference sees it."""
    import random

    from rgb_experiment_amd.utils import mask as M
    M._py_shuffled_range(1000, 7)
    after_native = random.getstate()
    random.seed(7)
    assert after_native == random.getstate()  # seeded, not advanced
    ref = list(range(1000))
    random.shuffle(ref)
    assert random.getstate() != after_native  # what CPython's own shuffle would have left


def test_edge_edits_refuse_out_of_range_endpoints_on_the_cpu_too():
    """coalesce / to_undirected: an endpoint outside [0, N) is an error on the CPU path as on the device path (one behaviour
    wherever the tensor lives; the reference's helpers alias such keys silently)."""
    import pytest
    import torch

    from rgb_experiment_amd.utils import coalesce, to_undirected
    ei = torch.tensor([[0, 1, 5], [1, 2, 0]])
    with pytest.raises(RuntimeError, match="1 endpoints outside"):
        coalesce(ei, 4)
    with pytest.raises(RuntimeError, match="outside"):
        to_undirected(torch.tensor([[0, -1], [1, 2]]), 4)
    assert to_undirected(ei, 6).size(1) == 6
