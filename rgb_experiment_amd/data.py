"""Minimal stand-in for torch_geometric.data.Data (reference itexperiments.py:23,188-189): a bag of
tensors with the handful of properties experiment() touches. No PyG dependency."""
import copy

import torch


class Data:
    def __init__(self, x=None, y=None, edge_index=None, **extra):
        self.x, self.y, self.edge_index = x, y, edge_index
        for k, v in extra.items():
            setattr(self, k, v)

    def keys(self):
        return [k for k, v in self.__dict__.items() if v is not None]

    @property
    def num_nodes(self):
        if self.x is not None:
            return self.x.size(0)
        if self.y is not None:
            return self.y.size(0)
        return int(self.edge_index.max()) + 1 if self.edge_index is not None and self.edge_index.numel() else 0

    @property
    def num_node_features(self):
        return 0 if self.x is None else (1 if self.x.dim() == 1 else self.x.size(1))

    @property
    def num_edges(self):
        return 0 if self.edge_index is None else self.edge_index.size(1)

    def _map(self, fn):
        out = Data()
        for k, v in self.__dict__.items():
            out.__dict__[k] = fn(v) if torch.is_tensor(v) else copy.deepcopy(v)
        return out

    def clone(self):
        return self._map(lambda t: t.clone())

    def to(self, device):
        return self._map(lambda t: t.to(device))

    def __repr__(self):
        parts = [f"{k}={list(v.shape) if torch.is_tensor(v) else v}" for k, v in self.__dict__.items()
                 if v is not None]
        return "Data(" + ", ".join(parts) + ")"
