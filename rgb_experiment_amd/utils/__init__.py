from .mask import get_whole_mask, get_classification_mask, get_random_mask, get_order, check_train_containing
from .subgraph import node_induced_subgraph
from .edges import to_undirected, coalesce, remove_self_loops, add_remaining_self_loops
