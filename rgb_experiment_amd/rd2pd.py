"""RD2PD — raw ``x.npy / y.npy / edge_index.npy`` triple -> Data with split masks (reference
rd2pd.py:20-148). One-shot, in front of the path; kept minimal so ``experiment(dataset_name=...)`` works
without PyG. Labels of -1 mark unlabelled nodes. ``device`` (an addition; the reference does all of this on the CPU,
rd2pd.py:3): the edge-list edits (:92-101) run on that device — coalesce in the HIP library (csrc/ingest.hip) — and the
returned Data lives there; masks are made on the CPU from the labels either way (bit-exact splitters, golden G4)."""
import os
import random

import numpy as np
import torch

from .data import Data
from .utils import (add_remaining_self_loops, coalesce, get_classification_mask, get_random_mask,
                    get_whole_mask, node_induced_subgraph, remove_self_loops)


class RD2PD:
    def __init__(self, dataset_name, dataset_root, *, split_method="ratio", split_seed=1234567,
                 split_ratio="6-2-2", num_train_per_class=20, num_val=500, num_test=1000,
                 remove_duplicate_edges=False, remove_self_loop=False, add_remaining_self_loop=False,
                 remove_non_label_node=False, specify_non_label_mask=False, apply_sample=False,
                 sample_seed=1234567, sample_method="random", sample_criterion="node",
                 sample_count_method="ratio", sample_num=1000, sample_ratio=0.8, sample_rw_length=None, device=None):
        folder = os.path.join(dataset_root, dataset_name)
        x = torch.from_numpy(np.load(os.path.join(folder, "x.npy"))).to(torch.float)
        y = torch.from_numpy(np.load(os.path.join(folder, "y.npy"))).to(torch.long)
        edge_index = torch.from_numpy(np.load(os.path.join(folder, "edge_index.npy"))).to(torch.long)
        self.num_nodes = x.size(0)
        if device is not None:
            edge_index = edge_index.to(device)

        # rd2pd.py:92-101
        if remove_duplicate_edges:
            edge_index = coalesce(edge_index, self.num_nodes)
        if remove_self_loop:
            edge_index = remove_self_loops(edge_index)
        if add_remaining_self_loop:
            edge_index = add_remaining_self_loops(edge_index, self.num_nodes)

        if device is not None and (remove_non_label_node or apply_sample):
            edge_index = edge_index.cpu()  # node_induced_subgraph is host logic (golden G5)

        # rd2pd.py:104-109
        if remove_non_label_node:
            keep = y != -1
            edge_index = node_induced_subgraph(self.num_nodes, keep, edge_index)
            x, y = x[keep], y[keep]
            self.num_nodes = x.size(0)

        # rd2pd.py:112-124: node-induced random sample
        if apply_sample and sample_method == "random" and sample_criterion == "node" \
                and sample_count_method == "ratio":
            random.seed(sample_seed)
            k = int(self.num_nodes * sample_ratio)
            picked = random.sample(list(range(self.num_nodes)), k)
            edge_index = node_induced_subgraph(self.num_nodes, picked, edge_index)
            x, y = x[picked], y[picked]
            self.num_nodes = k

        data = Data(x=x, y=y, edge_index=edge_index)
        self.non_label_mask = y == -1
        if specify_non_label_mask:
            data.non_label_mask = self.non_label_mask

        if split_method == "ratio":
            masks = get_whole_mask(y, split_ratio, split_seed)
        elif split_method == "classification":
            masks = get_classification_mask(y, split_ratio, split_seed)
        elif split_method == "random":
            masks = get_random_mask(y, num_train_per_class, num_val, num_test, split_seed)
        else:
            raise ValueError(f"split_method {split_method!r} not in ratio/classification/random")
        data.train_mask, data.val_mask, data.test_mask = masks
        self.data = data if device is None else data.to(device)
