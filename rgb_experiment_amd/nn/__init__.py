from .conv import GCNConv, SAGEConv, MySAGEConv, GATConv, APPNP, SGConv, GINConv
from .batchnorm import BatchNorm1d
from .linear import Linear
from .correct_and_smooth import CorrectAndSmooth, LabelPropagation

__all__ = ["GCNConv", "SAGEConv", "MySAGEConv", "GATConv", "APPNP", "SGConv", "GINConv",
           "CorrectAndSmooth", "LabelPropagation", "BatchNorm1d", "Linear"]
