"""More ranks than GPUs under RCCL (rehearsals on a box with fewer GPUs than ranks — a ONE-GPU box above all).

RCCL refuses two ranks of one host on the same device ("Duplicate GPU detected", ncclInvalidUsage). The check compares the
host identity of the ranks; NCCL_HOSTID replaces it. With a different NCCL_HOSTID per rank the ranks look like separate hosts,
the check passes, and RCCL connects them through its socket transport on the loopback interface (GPU -> host buffer ->
socket -> host buffer -> GPU). Nothing about the RATE of such a run means anything — the ranks also time-slice one GPU's CUs and
HBM — but everything about its BEHAVIOUR is the product's: torch.distributed's nccl backend with more than one rank,
communicator set-up, grouped send / recv lists on views (Comm.all_to_all_views), async works on RCCL's stream waited for out of
order, two host threads issuing in turns (TakeTurns), the side streams of the interleaved eval forwards. tests/test_gpu_dist.py
runs on it; bench.py --gpus N and experiment() take it when two of their ranks sit on the same device and say so on their output.

Whether ranks share a device is decided by what they actually opened, not by counting: every rank publishes the identity of its
device (host name + PCI domain:bus:device) through the rendezvous store BEFORE the communicator exists, and only when two ranks
name the same device do the ranks take the socket route. A launcher that gives each rank ONE visible GPU (HIP_VISIBLE_DEVICES
per rank: eight ranks, device_count() == 1 everywhere) therefore stays on xGMI."""
import os


def rccl_env(rank):
    """Variables a rank sets BEFORE the communicator is created (the NCCL_* variables are read at RCCL's init, which is later
    than this process's first GPU call). HSA_ENABLE_IPC_MODE_LEGACY is NOT among them: the HSA runtime reads it when it starts,
    i.e. before anything here can run — the launcher's environment has to carry it (bench.py sets it before `import torch`,
    the image exports it; a re-exec of a process that has touched the GPU is not an option on this pool)."""
    return {"NCCL_HOSTID": f"rgbx-shared-device-rank{rank}", "NCCL_SOCKET_IFNAME": "lo", "NCCL_IB_DISABLE": "1"}


def share_a_device(identities):
    """True when two ranks named the same device."""
    return len(set(identities)) < len(identities)


def apply_env(rank):
    """rccl_env(rank) into this process's environment; values the caller exported stay."""
    for k, v in rccl_env(rank).items():
        os.environ.setdefault(k, v)


def device_identity(device):
    import socket

    import torch
    p = torch.cuda.get_device_properties(device)
    return f"{socket.gethostname()}|{p.pci_domain_id:x}:{p.pci_bus_id:x}:{p.pci_device_id:x}"


def rendezvous_store(timeout_s=600):
    """(store, rank, world) of the launcher's environment (env://: a torchrun agent's store when there is one), BEFORE any
    process group exists."""
    import datetime

    from torch.distributed import rendezvous
    timeout = datetime.timedelta(seconds=timeout_s)
    store, rank, world = next(iter(rendezvous("env://", int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]),
                                              timeout=timeout)))
    store.set_timeout(timeout)
    return store, rank, world


def exchange_identities(store, rank, world, identity):
    store.set(f"rgbx/device/{rank}", identity)
    return [store.get(f"rgbx/device/{r}").decode() for r in range(world)]  # blocks until rank r has published


def init_rccl(device, timeout_s=600):
    """dist.init_process_group("nccl", device_id=device) from the launcher's environment (RANK / WORLD_SIZE / MASTER_*), with the
    device identities exchanged through the rendezvous store first (module docstring). Returns whether ranks share a device."""
    import datetime

    import torch.distributed as dist
    store, rank, world = rendezvous_store(timeout_s)
    shared = share_a_device(exchange_identities(store, rank, world, device_identity(device)))
    if shared:
        apply_env(rank)
    dist.init_process_group("nccl", store=store, rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=timeout_s), device_id=device)
    return shared
