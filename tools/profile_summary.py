#!/usr/bin/env python3
"""Turn rocprofv3 output directories into the summaries committed under profiles/.

  stats   <dir> --out profiles/X_kernel_stats.csv --cmd "<the profiled command>" [--top 40]
      copies rocprofv3's *_kernel_stats.csv (from `rocprofv3 --kernel-trace --stats -d <dir> -o <p> --output-format csv
      -- python3 bench.py ...`: the program itself after `--`, no wrapper hop) under a header naming the command.
  pmc     <dir> [<dir> ...] --out profiles/X_pmc.csv --cmd "..."
      per (kernel, counter): dispatches, mean, min, max over the *_counter_collection.csv of each PMC pass
      (FETCH_SIZE and WRITE_SIZE need separate passes: the TCC has 4 counter slots, FETCH_SIZE takes 3).
  traffic --fetch <dir> --write <dir> --kernel <substring> --workload L --model gcn [--out profiles/...json]
      HBM-side bytes per launch of one kernel = FETCH_SIZE x 2 (gfx950: wide reads are tallied at half their size,
      MI355X_MICROARCH.md "HBM") + WRITE_SIZE, both reported in KiB; stamped with the hash of the kernel's source
      files so that bench.py can tell a stale figure (bench.kernel_source_hash).
"""
import argparse
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _one(d, pattern):
    hits = sorted(glob.glob(os.path.join(d, "**", pattern), recursive=True))
    if not hits:
        raise SystemExit(f"no {pattern} under {d}")
    return hits[-1]


def counters(d):
    """{(kernel, counter): [values per dispatch]} of one rocprofv3 --pmc output directory."""
    out = defaultdict(list)
    with open(_one(d, "*counter_collection.csv"), newline="") as f:
        for row in csv.DictReader(f):
            out[(row["Kernel_Name"], row["Counter_Name"])].append(float(row["Counter_Value"]))
    return out


def cmd_stats(a):
    src = _one(a.dir, "*kernel_stats.csv")
    with open(src) as f:
        lines = f.read().splitlines()
    with open(a.out, "w") as f:
        f.write(f"# {a.cmd}\n")
        f.write("\n".join(lines[:a.top + 1]) + "\n")
    print(f"wrote {a.out} ({min(len(lines) - 1, a.top)} kernels)")


def cmd_pmc(a):
    rows = []
    for d in a.dirs:
        for (kernel, counter), vals in sorted(counters(d).items()):
            rows.append((kernel[:110], counter, len(vals), sum(vals) / len(vals), min(vals), max(vals)))
    rows.sort(key=lambda r: (r[1], -r[3] * r[2]))
    with open(a.out, "w") as f:
        f.write(f"# {a.cmd}\n# FETCH_SIZE / WRITE_SIZE in KiB as reported; gfx950: double FETCH_SIZE for wide coalesced reads\n")
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "dispatches", "avg", "min", "max"])
        for r in rows[:a.top]:
            w.writerow([r[0], r[1], r[2], f"{r[3]:.1f}", f"{r[4]:.1f}", f"{r[5]:.1f}"])
    print(f"wrote {a.out} ({min(len(rows), a.top)} rows)")


def cmd_traffic(a):
    import bench

    def mean_of(d, counter):
        vals = [v for (k, c), vs in counters(d).items() if c == counter and a.kernel in k for v in vs]
        if not vals:
            raise SystemExit(f"no {counter} dispatches of a kernel matching {a.kernel!r} under {d}")
        return sum(vals) / len(vals), len(vals)

    fetch, nf = mean_of(a.fetch, "FETCH_SIZE")
    write, nw = mean_of(a.write, "WRITE_SIZE")
    base = a.kernel.split("<")[0]
    rec = {"workload": a.workload, "model": a.model, "kernel": base, "kernel_match": a.kernel,
           "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write, "fetch_correction": 2.0,
           "dispatches": {"fetch_pass": nf, "write_pass": nw},
           "traffic_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
           "source_hash": bench.kernel_source_hash(base), "measured": a.measured,
           "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of the same bench.py "
                  "command; FETCH_SIZE doubled (gfx950 tallies wide reads at half their size); mean over the dispatches"}
    out = a.out or os.path.join(ROOT, "profiles", f"pmc_traffic_{a.workload}_{a.model}.json")
    with open(out, "w") as f:
        json.dump(rec, f, indent=1)
    print(f"wrote {out}: {rec['traffic_bytes_per_launch'] / 1e9:.3f} GB per launch")


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="what", required=True)
    s = sub.add_parser("stats")
    s.add_argument("dir")
    s.add_argument("--out", required=True)
    s.add_argument("--cmd", default="")
    s.add_argument("--top", type=int, default=40)
    s.set_defaults(fn=cmd_stats)
    p = sub.add_parser("pmc")
    p.add_argument("dirs", nargs="+")
    p.add_argument("--out", required=True)
    p.add_argument("--cmd", default="")
    p.add_argument("--top", type=int, default=60)
    p.set_defaults(fn=cmd_pmc)
    t = sub.add_parser("traffic")
    t.add_argument("--fetch", required=True)
    t.add_argument("--write", required=True)
    t.add_argument("--kernel", required=True)
    t.add_argument("--workload", required=True)
    t.add_argument("--model", required=True)
    t.add_argument("--measured", default="")
    t.add_argument("--out")
    t.set_defaults(fn=cmd_traffic)
    a = ap.parse_args()
    a.fn(a)


if __name__ == "__main__":
    main()
