#!/usr/bin/env python3
"""Per-epoch kernel breakdown from two `rocprofv3 --kernel-trace --stats` runs of the same bench.py command that differ
only in --steps: everything outside the timed epochs (set-up, warm-up, CSR builds) cancels in the difference.
Usage: python tools/epoch_diff.py <dir A> <steps A> <dir B> <steps B> [--out file] [--top N]"""
import csv
import glob
import sys


def load(d):
    f = sorted(glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True))[-1]
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}


def main():
    da, sa, db, sb = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 45
    a, b = load(da), load(db)
    n = sb - sa
    rows = []
    for k, (c2, t2) in b.items():
        c1, t1 = a.get(k, (0, 0.0))
        if c2 > c1:
            rows.append((k, (c2 - c1) / n, (t2 - t1) / n / 1e6))
    rows.sort(key=lambda r: -r[2])
    lines = [f"# per epoch: difference of {db} ({sb} steps) and {da} ({sa} steps)",
             f"# GPU ms per epoch {sum(r[2] for r in rows):.3f} in {sum(r[1] for r in rows):.1f} launches",
             "kernel,launches_per_epoch,ms_per_epoch"]
    lines += [f"\"{k[:140]}\",{c:.1f},{t:.4f}" for k, c, t in rows[:top]]
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
