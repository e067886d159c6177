"""More ranks than GPUs under RCCL (rehearsals on a box with fewer GPUs than ranks — a ONE-GPU box above all).

RCCL refuses two ranks of one host on the same device ("Duplicate GPU detected", ncclInvalidUsage). The check compares the
host identity of the ranks; NCCL_HOSTID replaces it. With a different NCCL_HOSTID per rank the ranks look like separate hosts,
the check passes, and RCCL connects them through its socket transport on the loopback interface (GPU -> host buffer ->
socket -> host buffer -> GPU). Nothing about the RATE of such a run means anything — the ranks also time-slice one GPU's CUs and
HBM — but everything about its BEHAVIOUR is the product's: torch.distributed's nccl backend with more than one rank,
communicator set-up, grouped send / recv lists on views (Comm.all_to_all_views), async works on RCCL's stream waited for out of
order, two host threads issuing in turns (TakeTurns), the side streams of the interleaved eval forwards. tests/test_gpu_dist.py
runs on it; bench.py --gpus N and experiment() take it when N exceeds the visible devices and say so on their output."""
import os


def ranks_share_devices(world, n_devices):
    return world > max(int(n_devices), 1)


def rccl_env(rank):
    """Variables a rank sets BEFORE the communicator is created (they are read at RCCL's init)."""
    return {"NCCL_HOSTID": f"rgbx-shared-device-rank{rank}", "NCCL_SOCKET_IFNAME": "lo", "NCCL_IB_DISABLE": "1",
            "HSA_ENABLE_IPC_MODE_LEGACY": "0"}


def prepare_rccl(rank, world, n_devices):
    """Called by every rank before dist.init_process_group("nccl"): when the ranks must share devices, put rccl_env()
    into this process's environment (values the caller exported stay). Returns whether the ranks share devices."""
    if not ranks_share_devices(world, n_devices):
        return False
    for k, v in rccl_env(rank).items():
        os.environ.setdefault(k, v)
    return True
