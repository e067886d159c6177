"""Multi-GPU full-graph training: 1-D node partition; per propagate an RCCL all-to-all under the halo,
transpose or row-group x column-slice scheme — or no activation exchange at all (ReplicaGraph) — chosen by a cost model."""
from .comm import Comm, EmulatedComm
from .graph import DistGraph, HipAggregator, ReplicaGraph, install, install_replicas
from .nn import DistBatchNorm1d
from .plan import GridPlan, HalfPlan, PartitionPlan, partition_bounds
from .runner import DistRunner
from .tasksplit import TaskSplitRunner

DistGCNRunner = DistRunner  # bench.py's name for the 2-layer GCN workload

__all__ = ["Comm", "EmulatedComm", "DistGraph", "ReplicaGraph", "HipAggregator", "install", "install_replicas",
           "DistBatchNorm1d", "GridPlan", "HalfPlan", "PartitionPlan", "partition_bounds", "DistRunner", "DistGCNRunner"]
