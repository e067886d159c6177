#!/usr/bin/env python3
"""Host time to ISSUE one exchange on torch.distributed's nccl backend (RCCL), by form — the cost a rank's host thread pays per
exchange while its GPU queue must stay fed (an emulated rank pays none: EmulatedComm). Ranks share the one GPU (dist/sharing.py);
the payloads are small so that the socket transport is not what is measured. Forms: grouped send / recv lists
(dist.all_to_all, what Comm.all_to_all_views issues), all_to_all_single with split sizes (Comm.all_to_all_rows), an async
all_reduce, and batch_isend_irecv. Printed per form: median microseconds of the call itself (async_op=True, returns a work) and of
call + work.wait() (the stream-side wait, no host sync).
Usage: python tools/rccl_issue_cost.py [world=4] [iterations=300]"""
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import torch
    import torch.distributed as dist
    from rgb_experiment_amd.dist import sharing
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    iters = int(os.environ["RGBX_ITERS"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    shared = sharing.init_rccl(dev)
    rows, w = 64, 32
    send_buf = torch.randn(world, rows, w, device=dev)
    recv_buf = torch.empty(world, rows, w, device=dev)
    red = torch.randn(256, device=dev)

    def lists():
        return dist.all_to_all([recv_buf[q] for q in range(world)], [send_buf[q] for q in range(world)], async_op=True)

    def single():
        return dist.all_to_all_single(recv_buf.view(-1, w), send_buf.view(-1, w), [rows] * world, [rows] * world, async_op=True)

    def reduce():
        return dist.all_reduce(red, async_op=True)

    def p2p():
        ops = []
        for q in range(world):
            if q != rank:
                ops.append(dist.P2POp(dist.isend, send_buf[q], q))
                ops.append(dist.P2POp(dist.irecv, recv_buf[q], q))
        works = dist.batch_isend_irecv(ops)

        class _All:
            def wait(self):
                for x in works:
                    x.wait()
        return _All()

    out = {}
    for name, fn in (("all_to_all lists", lists), ("all_to_all_single", single), ("all_reduce async", reduce),
                     ("batch_isend_irecv", p2p)):
        for _ in range(20):
            fn().wait()
        torch.cuda.synchronize()
        dist.barrier()
        issue, both = [], []
        for i in range(iters):
            t0 = time.perf_counter()
            work = fn()
            t1 = time.perf_counter()
            work.wait()
            t2 = time.perf_counter()
            issue.append((t1 - t0) * 1e6)
            both.append((t2 - t0) * 1e6)
            if i % 16 == 15:
                torch.cuda.synchronize()  # keep the queue short: issue cost, not back-pressure
        torch.cuda.synchronize()
        out[name] = (statistics.median(issue), statistics.median(both))
    if rank == 0:
        print(f"world {world}, ranks share a device: {shared}; host microseconds per exchange (median of {iters})")
        for name, (a, b) in out.items():
            print(f"  {name:22s} issue {a:7.1f} us   issue + work.wait() {b:7.1f} us", flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    iters = sys.argv[2] if len(sys.argv) > 2 else "300"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), RGBX_ISSUE_CHILD="1", RGBX_ITERS=iters, OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env))
    rc = 0
    for p in procs:
        try:
            rc |= p.wait(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill()
            rc |= 1
    return rc


if __name__ == "__main__":
    if os.environ.get("RGBX_ISSUE_CHILD"):
        child()
    else:
        sys.exit(main())
