from .conv import GCNConv, SAGEConv, MySAGEConv, GATConv, APPNP

__all__ = ["GCNConv", "SAGEConv", "MySAGEConv", "GATConv", "APPNP"]
