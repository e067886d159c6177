from .conv import GCNConv, SAGEConv, MySAGEConv, GATConv, APPNP, SGConv, GINConv

__all__ = ["GCNConv", "SAGEConv", "MySAGEConv", "GATConv", "APPNP", "SGConv", "GINConv"]
