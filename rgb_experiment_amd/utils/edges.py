"""Host-side edge-list utilities with the semantics of the PyG helpers the reference calls once
per run, outside the hot loop: to_undirected (itexperiments.py:238), coalesce /
remove_self_loops / add_remaining_self_loops (rd2pd.py:92-101) [PyG]."""
import torch


def coalesce(edge_index, num_nodes):
    """Sort edges by (row, col) and drop duplicates."""
    if edge_index.numel() == 0:
        return edge_index
    key = edge_index[0] * num_nodes + edge_index[1]
    key = torch.unique(key, sorted=True)
    return torch.stack([torch.div(key, num_nodes, rounding_mode="floor"), key % num_nodes])


def to_undirected(edge_index, num_nodes=None):
    """Add every reverse edge, then coalesce."""
    if num_nodes is None:
        num_nodes = int(edge_index.max()) + 1 if edge_index.numel() else 0
    both = torch.cat([edge_index, edge_index.flip(0)], dim=1)
    return coalesce(both, num_nodes)


def remove_self_loops(edge_index):
    return edge_index[:, edge_index[0] != edge_index[1]]


def add_remaining_self_loops(edge_index, num_nodes):
    """Non-loop edges in order, followed by one self-loop per node."""
    loops = torch.arange(num_nodes, dtype=edge_index.dtype, device=edge_index.device)
    return torch.cat([remove_self_loops(edge_index), loops.unsqueeze(0).repeat(2, 1)], dim=1)
