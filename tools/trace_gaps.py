#!/usr/bin/env python3
"""Idle time between consecutive kernels of the last epochs in a rocprofv3 --kernel-trace CSV.
Usage: python tools/trace_gaps.py <p_kernel_trace.csv> <launches per epoch> [epochs]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = int(float(sys.argv[2]))
epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 3
tail = rows[-per * epochs:]
for e in range(epochs):
    chunk = tail[e * per:(e + 1) * per]
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in chunk) / 1e3
    span = (int(chunk[-1]["End_Timestamp"]) - int(chunk[0]["Start_Timestamp"])) / 1e3
    gaps = []
    prev = None
    for r in chunk:
        if prev is not None:
            gaps.append(((int(r["Start_Timestamp"]) - prev) / 1e3, r["Kernel_Name"][:60]))
        prev = int(r["End_Timestamp"])
    big = sorted(gaps, reverse=True)[:6]
    print(f"epoch -{epochs - e}: span {span:.0f} us, busy {busy:.0f} us, idle {span - busy:.0f} us in {len(gaps)} gaps; "
          f"gaps > 4 us: {sum(1 for g, _ in gaps if g > 4)}; largest: " + "; ".join(f"{g:.0f} us before {k}" for g, k in big))
