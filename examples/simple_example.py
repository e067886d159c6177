#!/usr/bin/env python3
"""Counterpart of the reference's examples/simple_example.py (and simple_cs_example.py): experiment()
loads `<root>/<name>/{x,y,edge_index}.npy` through RD2PD, trains one model by `model_name`, prints the
metric dict. The reference's datasets are not shipped, so a Cora-shaped synthetic graph is written first.

    python examples/simple_example.py [model_name]        # gcn | graphsage | graphsage2 | gat | appnpstack | ...
"""
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgb_experiment_amd import InitialParameters, experiment  # noqa: E402


def write_cora_shaped(root, name="cora_like", n=2708, pairs=5278, f=1433, c=7, seed=0):
    g = torch.Generator().manual_seed(seed)
    y = torch.randint(0, c, (n,), generator=g)
    a = torch.randint(0, n, (pairs,), generator=g)
    # homophilous: most edges stay inside a class
    same = torch.rand(pairs, generator=g) < 0.8
    cand = torch.randint(0, n, (pairs, 16), generator=g)
    match = (y[cand] == y[a].unsqueeze(1)).float()
    pick = torch.where(same, match.argmax(dim=1), torch.zeros(pairs, dtype=torch.long))
    b = cand[torch.arange(pairs), pick]
    keep = a != b
    a, b = a[keep], b[keep]
    x = torch.zeros(n, f)
    words = torch.randint(0, f // c, (n, 18), generator=g) + (y * (f // c)).unsqueeze(1) * (torch.rand(n, 18, generator=g) < 0.6)
    x.scatter_(1, words.clamp(max=f - 1), 1.0)
    folder = os.path.join(root, name)
    os.makedirs(folder, exist_ok=True)
    np.save(os.path.join(folder, "x.npy"), x.numpy())
    np.save(os.path.join(folder, "y.npy"), y.numpy())
    np.save(os.path.join(folder, "edge_index.npy"), torch.cat([torch.stack([a, b]), torch.stack([b, a])], 1).numpy())
    return name


if __name__ == "__main__":
    model_name = sys.argv[1] if len(sys.argv) > 1 else "gcn"
    with tempfile.TemporaryDirectory() as root:
        dataset_name = write_cora_shaped(root)
        acc_dict = experiment(model_init_param=InitialParameters.defaults_for(model_name), dataset_name=dataset_name,
                              dataset_root=root, dataset_split_mode="ratio", model_name=model_name,
                              dataset_split_seed=14530529, learning_rate=0.01, epoch=100, normalize_feature="row",
                              post_cs=("--cs" in sys.argv), cs_param=InitialParameters.default_cs_param)
    print(acc_dict)
