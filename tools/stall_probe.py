#!/usr/bin/env python3
"""Where do the occasional 80-140 ms steps of bench.py's timed loops come from (one step in ~50 on the pool's boxes)?
Runs the headline epoch (GCN, workload L) for many steps and, per step, reads the cgroup's CPU-throttling counters
(/sys/fs/cgroup/cpu.stat nr_throttled / throttled_usec), Python's GC counts and the process's thread count; prints every step
beyond 1.5 x the median with the deltas. Usage: python tools/stall_probe.py [steps] [torch threads]"""
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
threads = int(sys.argv[2]) if len(sys.argv) > 2 else bench.host_cores()
torch.set_num_threads(threads)


def cpu_stat():
    out = {}
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            k, v = line.split()
            out[k] = int(v)
    except OSError:
        pass
    return out


dev = torch.device("cuda", 0)
wl = bench.WORKLOADS["L"]
N, E, d = wl["N"], wl["E"], wl["d"]
ei, x, y = bench.synth(N, E, d)
masks = bench.split_masks(y)
kwargs, n_prop, loops_mode, kind = bench.MODELS["gcn"]
torch.manual_seed(14530529)
model = bench.model_class("gcn")(input_dim=d, output_dim=d, **kwargs)
step, nnz, _ = bench.build_single_gpu(model, ei, x, y, masks, dev, loops_mode, kind, N, d)
for _ in range(5):
    step()
torch.cuda.synchronize()
print(f"torch threads {torch.get_num_threads()}, process threads {len(os.listdir('/proc/self/task'))}, gc thresholds {gc.get_threshold()}",
      flush=True)
rows = []
t_prev, s_prev, g_prev = time.perf_counter(), cpu_stat(), gc.get_stats()
for i in range(steps):
    step()
    t, s, g = time.perf_counter(), cpu_stat(), gc.get_stats()
    rows.append(((t - t_prev) * 1e3, s.get("nr_throttled", 0) - s_prev.get("nr_throttled", 0),
                 (s.get("throttled_usec", 0) - s_prev.get("throttled_usec", 0)) / 1e3,
                 (s.get("usage_usec", 0) - s_prev.get("usage_usec", 0)) / 1e3,
                 [b["collections"] - a["collections"] for a, b in zip(g_prev, g)]))
    t_prev, s_prev, g_prev = t, s, g
ms = sorted(r[0] for r in rows)
med = ms[len(ms) // 2]
print(f"{steps} steps: median {med:.2f} ms, max {ms[-1]:.2f} ms, mean {sum(ms) / len(ms):.2f} ms; cgroup CPU per step (median) "
      f"{sorted(r[3] for r in rows)[len(rows) // 2]:.1f} ms")
for i, r in enumerate(rows):
    if r[0] > 1.5 * med:
        print(f"step {i}: {r[0]:.1f} ms   throttled periods +{r[1]}  throttled time +{r[2]:.1f} ms  cgroup CPU {r[3]:.1f} ms  gc collections {r[4]}")
