"""Node-induced subgraph relabelling (reference utils/subgraph.py:6-32); host-side, used by RD2PD."""
import torch


def node_induced_subgraph(total_num_nodes, nodes_set, original_edge_index, reorder_nodes=True):
    """Keep the edges whose two endpoints are in `nodes_set` (index list or bool mask); with
    `reorder_nodes` relabel the kept nodes 0..k-1 in `nodes_set` order."""
    if isinstance(nodes_set, torch.Tensor) and nodes_set.dtype == torch.bool:
        kept = int(nodes_set.sum())
    else:
        kept = len(nodes_set)
    chosen = torch.zeros(total_num_nodes, dtype=torch.bool)
    chosen[nodes_set] = True
    src, dst = original_edge_index[0], original_edge_index[1]
    edge_index = original_edge_index[:, chosen[src] & chosen[dst]]
    if reorder_nodes:
        relabel = torch.zeros(total_num_nodes, dtype=torch.long)
        relabel[nodes_set] = torch.arange(kept)
        edge_index = relabel[edge_index]
    return edge_index
