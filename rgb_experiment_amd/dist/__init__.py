"""Multi-GPU full-graph training: 1-D node partition + RCCL all-to-all halo exchange."""
from .comm import Comm
from .graph import DistGraph, HipAggregator, install
from .nn import DistBatchNorm1d
from .plan import HalfPlan, PartitionPlan, partition_bounds
from .runner import DistRunner

DistGCNRunner = DistRunner  # bench.py's name for the 2-layer GCN workload

__all__ = ["Comm", "DistGraph", "HipAggregator", "install", "DistBatchNorm1d", "HalfPlan", "PartitionPlan",
           "partition_bounds", "DistRunner", "DistGCNRunner"]
