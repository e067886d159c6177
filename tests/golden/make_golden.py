"""Generate the golden fixtures under tests/golden/ from the REFERENCE's own PyG-free code.

Runs only in the build container (needs /root/reference; the GPU box never sees it). The
reference cannot be imported as-is: ``rgb_experiment/__init__.py`` pulls in torch_geometric,
torch_scatter and torch_sparse (itexperiments.py:21-23, models/dagnn.py:10, rd2pd.py:12), none of
which is installed (ordinary ModuleNotFoundError, not a permission denial). This script registers
INERT placeholder modules for exactly those three package names — empty module objects whose
attributes are empty placeholder classes, holding no arithmetic — so that the import statement
succeeds, and then executes ONLY functions that never touch a placeholder:

  G1  itexperiments.edge_index2sparse_matrix + normalize_adj   (itexperiments.py:671-684)
  G2  itexperiments.label_propagation                          (itexperiments.py:698-719)
  G3  models.pta.PTA.inference                                 (models/pta.py:79-84)
  G4  utils.mask.get_whole_mask / get_classification_mask / get_random_mask
  G5  utils.subgraph.node_induced_subgraph
  G6  itexperiments.compare_pred_label (need_all_metrics=True)
  G7  models.pta.PTA.forward / loss_function (train mode 0/1/2 at several epochs, eval mode)

Outputs are inputs + expected outputs only (``*.npz``); no reference source text is stored.
Usage:  python tests/golden/make_golden.py
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types
import warnings

import numpy as np
import scipy.sparse as sp
import torch

REF_ROOT = "/root/reference"
OUT_DIR = os.path.dirname(os.path.abspath(__file__))
_PLACEHOLDER_ROOTS = ("torch_geometric", "torch_scatter", "torch_sparse")


class _Inert(types.ModuleType):
    """Module whose every attribute is an empty class (so `class X(MessagePassing)` parses)."""
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (), {"__init__": lambda self, *a, **k: None})
        setattr(self, name, cls)
        return cls


class _InertFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in _PLACEHOLDER_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _Inert(spec.name)

    def exec_module(self, module):
        pass


def load_reference():
    sys.meta_path.insert(0, _InertFinder())
    sys.path.insert(0, REF_ROOT)
    import rgb_experiment.itexperiments as it
    import rgb_experiment.models.pta as pta
    import rgb_experiment.utils.mask as mask
    import rgb_experiment.utils.subgraph as subgraph
    return it, pta, mask, subgraph


GRAPHS = {
    # name: (num_nodes, edge_index) — all WITHOUT self-loops (normalize_adj adds I itself)
    "survey4": (4, [[0, 1, 1, 2, 3], [1, 0, 2, 1, 0]]),                  # SURVEY §8c sample, directed
    "path5_undirected": (5, [[0, 1, 1, 2, 2, 3, 3, 4], [1, 0, 2, 1, 3, 2, 4, 3]]),
    "star6_undirected": (6, [[0, 0, 0, 0, 0, 1, 2, 3, 4, 5], [1, 2, 3, 4, 5, 0, 0, 0, 0, 0]]),
    "directed7_isolated": (7, [[0, 0, 1, 2, 4, 4, 5], [1, 2, 2, 0, 5, 0, 4]]),  # node 3 and 6 isolated
}


def _random_undirected(n, pairs, seed):
    g = torch.Generator().manual_seed(seed)
    a = torch.randint(0, n, (pairs,), generator=g)
    b = torch.randint(0, n, (pairs,), generator=g)
    keep = a != b
    a, b = a[keep], b[keep]
    key = torch.unique(torch.minimum(a, b) * n + torch.maximum(a, b))
    u, v = key // n, key % n
    return torch.stack([torch.cat([u, v]), torch.cat([v, u])])


def main():
    warnings.simplefilter("ignore")
    it, pta, mask, subgraph = load_reference()
    out = {}

    graphs = {k: (n, torch.tensor(ei, dtype=torch.int64)) for k, (n, ei) in GRAPHS.items()}
    graphs["random40_undirected"] = (40, _random_undirected(40, 90, 7))

    # ---- G1 / G2 / G3 ----------------------------------------------------------------------
    for name, (n, ei) in graphs.items():
        adj = it.edge_index2sparse_matrix(ei, n)
        adj = adj + sp.eye(adj.shape[0])
        adj = it.normalize_adj(adj)
        dense = np.asarray(adj.todense(), dtype=np.float64)
        out[f"g1/{name}/edge_index"] = ei.numpy()
        out[f"g1/{name}/num_nodes"] = np.int64(n)
        out[f"g1/{name}/adj_ref"] = dense          # adj[src, dst] convention of itexperiments.py:675

        adj_t = it.sparse_mx_to_torch_sparse_tensor(adj)
        g = torch.Generator().manual_seed(100 + n)
        labels = torch.randint(0, 3, (n,), generator=g)
        labels[0], labels[1], labels[2] = 0, 1, 2
        idx = torch.arange(0, n, 2)
        for K, alpha in ((3, 0.1), (10, 0.1)):
            y = it.label_propagation(adj_t, labels, idx, K, alpha, "cpu")
            out[f"g2/{name}/K{K}/out"] = y.numpy()
        out[f"g2/{name}/labels"] = labels.numpy()
        out[f"g2/{name}/idx"] = idx.numpy()

        h = torch.randn(n, 5, generator=g)
        for K, alpha in ((1, 0.1), (10, 0.1), (4, 0.35)):
            model = pta.PTA(nfeat=3, nhid=4, nclass=5, dropout=0.0, epsilon=100, K=K, alpha=alpha)
            y = model.inference(h, adj_t)
            out[f"g3/{name}/K{K}_a{alpha}/out"] = y.numpy()
        out[f"g3/{name}/h"] = h.numpy()

    # ---- G4 masks ----------------------------------------------------------------------------
    g = torch.Generator().manual_seed(4242)
    y_small = torch.randint(0, 4, (57,), generator=g)
    y_small[torch.tensor([3, 11, 40])] = -1
    y_cora = torch.randint(0, 7, (2708,), generator=torch.Generator().manual_seed(1234569))
    cases = {"small": y_small, "cora_shaped": y_cora}
    for cname, y in cases.items():
        out[f"g4/{cname}/y"] = y.numpy()
        for ratio, seed in (("6-2-2", 123456789), ("5-2-3", 14530529), ("1-1-3", 1234567)):
            m = mask.get_whole_mask(y, ratio, seed)
            out[f"g4/{cname}/whole/{ratio}/{seed}"] = torch.stack(m).numpy()
            m = mask.get_classification_mask(y, ratio, seed)
            out[f"g4/{cname}/classification/{ratio}/{seed}"] = torch.stack(m).numpy()
        m = mask.get_random_mask(y, 5, 10, 20, 1234567)
        out[f"g4/{cname}/random_train/5/1234567"] = m[0].numpy()  # only the train part is seed-pinned

    # ---- G5 node-induced subgraph -------------------------------------------------------------
    n, ei = graphs["random40_undirected"]
    picked = [5, 1, 9, 30, 31, 32, 2, 17, 18, 39, 0, 22]
    out["g5/edge_index"] = ei.numpy()
    out["g5/num_nodes"] = np.int64(n)
    out["g5/nodes_list"] = np.asarray(picked)
    out["g5/list_reorder"] = subgraph.node_induced_subgraph(n, picked, ei, True).numpy()
    out["g5/list_keep"] = subgraph.node_induced_subgraph(n, picked, ei, False).numpy()
    bmask = torch.zeros(n, dtype=torch.bool)
    bmask[torch.tensor(picked)] = True
    out["g5/nodes_mask"] = bmask.numpy()
    out["g5/mask_reorder"] = subgraph.node_induced_subgraph(n, bmask, ei, True).numpy()

    # ---- G6 metrics ---------------------------------------------------------------------------
    g = torch.Generator().manual_seed(66)
    label = torch.randint(0, 5, (200,), generator=g)
    pred = label.clone()
    flip = torch.rand(200, generator=g) < 0.3
    pred[flip] = torch.randint(0, 6, (int(flip.sum()),), generator=g)
    res = it.compare_pred_label(pred, label, True)
    out["g6/pred"], out["g6/label"] = pred.numpy(), label.numpy()
    out["g6/metrics"] = np.asarray([res["ACC"], res["precision_score"], res["recall_score"], res["f1_macro"],
                                    res["f1_micro"]], dtype=np.float64)

    # ---- G7 PTA forward + loss (pure torch; weights exported so the rebuild can load them) -------------
    torch.manual_seed(77)
    g = torch.Generator().manual_seed(78)
    xin = torch.randn(30, 6, generator=g)
    y_soft = torch.softmax(torch.randn(30, 4, generator=g), dim=-1)
    for mode in (0, 1, 2):
        model = pta.PTA(nfeat=6, nhid=5, nclass=4, dropout=0.0, epsilon=100, K=3, alpha=0.1, mode=mode)
        for k, v in model.state_dict().items():
            out[f"g7/mode{mode}/state/{k}"] = v.detach().numpy()
        model.train()
        y_hat = model(xin)
        out[f"g7/mode{mode}/forward"] = y_hat.detach().numpy()
        for epoch in (0, 7, 150):
            out[f"g7/mode{mode}/train_loss/{epoch}"] = model.loss_function(y_hat, y_soft, epoch).detach().numpy()
        model.eval()
        out[f"g7/mode{mode}/eval_loss"] = model.loss_function(model(xin), y_soft).detach().numpy()
    out["g7/x"], out["g7/y_soft"] = xin.numpy(), y_soft.numpy()

    path = os.path.join(OUT_DIR, "reference_pygfree.npz")
    np.savez_compressed(path, **{k.replace("/", "__"): v for k, v in out.items()})
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
