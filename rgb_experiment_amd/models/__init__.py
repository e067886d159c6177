"""Model registry (reference models/__init__.py:1-13): the hot-path models plus the "next" rows of
SURVEY §8f that reuse the same kernels (DAGNN, PTA, SGC, GIN). GGNN, SuperGAT and FAGCN are out of scope."""
from .mlp import MLP
from .gcn import GCN
from .graphsage import GraphSAGE
from .graphsage2 import GraphSAGE2
from .gat import GAT
from .appnp_stack import APPNPStack
from .dagnn import DAGNN
from .pta import PTA
from .sgc import SGC
from .gin import GIN

REGISTRY = {
    "mlp": MLP,
    "gcn": GCN,
    "graphsage": GraphSAGE,
    "graphsage2": GraphSAGE2,
    "gat": GAT,
    "appnpstack": APPNPStack,
    "dagnn": DAGNN,
    "pta": PTA,
    "sgc": SGC,
    "gin": GIN,
}
