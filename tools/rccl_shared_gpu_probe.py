#!/usr/bin/env python3
"""Can RCCL itself run with two ranks on a ONE-GPU box? RCCL refuses two ranks of one host on the same device ("Duplicate GPU
detected"); with a different NCCL_HOSTID per rank the ranks look like two hosts, the check passes and the bytes travel over the
socket transport on the loopback interface (GPU -> host -> socket -> host -> GPU). Nothing about the rate means anything; what it
exercises is the nccl backend of torch.distributed with more than one rank: communicator set-up, all_reduce, all_to_all_single on
device views, async work handles — the calls rgb_experiment_amd.dist makes.
Usage: python tools/rccl_shared_gpu_probe.py [world]   (parent; starts the ranks as children and never touches the GPU itself)"""
import os
import socket
import subprocess
import sys


def child():
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev)
    t = torch.full((1 << 20,), float(rank + 1), device=dev)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    assert t[0].item() == world * (world + 1) / 2, t[0].item()
    n = 1000
    send = (torch.arange(world * n, device=dev, dtype=torch.float32) + 1000 * rank).reshape(world * n // 4, 4)
    recv = torch.empty_like(send)
    work = dist.all_to_all_single(recv, send, async_op=True)
    work.wait()
    torch.cuda.synchronize()
    rows = n // 4
    for src in range(world):
        exp = (torch.arange(rank * n, (rank + 1) * n, device=dev, dtype=torch.float32) + 1000 * src).reshape(rows, 4)
        assert torch.equal(recv[src * rows:(src + 1) * rows], exp), (rank, src)
    # ragged splits, int64
    sizes_out = [(rank + p) % 3 + 1 for p in range(world)]
    sizes_in = [(p + rank) % 3 + 1 for p in range(world)]
    s = torch.arange(sum(sizes_out), device=dev) + 100 * rank
    r = torch.empty(sum(sizes_in), dtype=torch.int64, device=dev)
    dist.all_to_all_single(r, s, sizes_in, sizes_out)
    torch.cuda.synchronize()
    dist.barrier()
    if rank == 0 and os.environ.get("RGBX_PROBE_PROFILE"):
        # which GPU kernels carry the collectives (torch's own profiler, inside this process: no launcher to profile)
        from torch.profiler import ProfilerActivity, profile
        big = torch.randn(4 << 20, device=dev)
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            for _ in range(3):
                dist.all_reduce(big)
                dist.all_to_all_single(recv, send)
            torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=90), flush=True)
    elif os.environ.get("RGBX_PROBE_PROFILE"):
        big = torch.randn(4 << 20, device=dev)
        for _ in range(3):
            dist.all_reduce(big)
            dist.all_to_all_single(recv, send)
        torch.cuda.synchronize()
    print(f"rank {rank}/{world}: nccl backend ok (all_reduce, all_to_all_single even + ragged, barrier); "
          f"version {torch.cuda.nccl.version()}", flush=True)
    dist.destroy_process_group()


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), NCCL_HOSTID=f"rgbx-shared-gpu-rank{rank}", NCCL_SOCKET_IFNAME="lo",
                   NCCL_IB_DISABLE="1", NCCL_DEBUG=os.environ.get("NCCL_DEBUG", "WARN"), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   RGBX_PROBE_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env))
    rc = 0
    for p in procs:
        try:
            rc |= p.wait(timeout=150)
        except subprocess.TimeoutExpired:
            p.kill()
            rc |= 1
    print("probe", "ok" if rc == 0 else f"FAILED rc={rc}")
    return rc


if __name__ == "__main__":
    if os.environ.get("RGBX_PROBE_CHILD"):
        child()
    else:
        sys.exit(main())
