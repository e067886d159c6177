"""Model registry for the hot-path models (reference models/__init__.py:1-13). The other zoo
members (GGNN, PTA, DAGNN, SuperGAT, SGC, GIN, FAGCN) are out of scope."""
from .mlp import MLP
from .gcn import GCN
from .graphsage import GraphSAGE
from .graphsage2 import GraphSAGE2
from .gat import GAT
from .appnp_stack import APPNPStack

REGISTRY = {
    "mlp": MLP,
    "gcn": GCN,
    "graphsage": GraphSAGE,
    "graphsage2": GraphSAGE2,
    "gat": GAT,
    "appnpstack": APPNPStack,
}
