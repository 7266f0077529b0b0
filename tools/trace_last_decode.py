#!/usr/bin/env python3
"""Per-launch durations of the LAST decode in a rocprofv3 --kernel-trace run (tools/single_latency.py
under the profiler): the kernels after the last idle gap, in order, with the gaps between them.
  python tools/trace_last_decode.py DIR [--gap-us 300]"""
import argparse
import csv
import glob
import os

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--gap-us", type=float, default=300)
ap.add_argument("--which", type=int, default=-1, help="burst index (default: the last)")
a = ap.parse_args()
rows = []
for f in glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]))
rows.sort()
bursts, cur = [], [rows[0]]
for r in rows[1:]:
    if r[0] - cur[-1][1] > a.gap_us * 1e3:
        bursts.append(cur)
        cur = []
    cur.append(r)
bursts.append(cur)
b = bursts[a.which]
print(f"{len(bursts)} bursts; burst {a.which}: {len(b)} kernels over {(b[-1][1] - b[0][0]) / 1e3:.1f} us")
prev = None
for s, e, n in b:
    print(f"  {n:40s} {(e - s) / 1e3:8.1f} us   gap before {((s - prev) / 1e3) if prev else 0:7.1f} us")
    prev = e
