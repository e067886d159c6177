"""Host-side helpers around the hot path: split masks (bit-exact with the reference's seeded splitters, golden
G4), node-induced subgraphs (golden G5) and the edge-list edits in front of the path (utils/edges.py: on a device
tensor coalesce / to_undirected run in the HIP library, csrc/ingest.hip)."""
from . import edges, mask, subgraph

# split masks (reference utils/mask.py)
get_whole_mask = mask.get_whole_mask
get_classification_mask = mask.get_classification_mask
get_random_mask = mask.get_random_mask
get_order = mask.get_order
check_train_containing = mask.check_train_containing

# subgraphs (reference utils/subgraph.py)
node_induced_subgraph = subgraph.node_induced_subgraph

# edge-list edits used by RD2PD / experiment(to_undirected=...)
to_undirected = edges.to_undirected
coalesce = edges.coalesce
remove_self_loops = edges.remove_self_loops
add_remaining_self_loops = edges.add_remaining_self_loops

__all__ = ["get_whole_mask", "get_classification_mask", "get_random_mask", "get_order", "check_train_containing",
           "node_induced_subgraph", "to_undirected", "coalesce", "remove_self_loops", "add_remaining_self_loops"]
