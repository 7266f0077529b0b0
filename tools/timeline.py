#!/usr/bin/env python3
"""What the device and the link were doing during the LAST batch of a traced run:

    rocprofv3 --kernel-trace --memory-copy-trace -d DIR -o NAME --output-format csv -- python3 tools/e2e_bench.py ...
    python tools/timeline.py DIR

reads the kernel and memory-copy traces, takes the last burst of activity (activity separated
from what came before by an idle gap of --gap ms: the last timed pass of e2e_bench.py), and prints,
per lane of activity (host->device copies, device->host copies, each kernel), the time it was busy
(union of its intervals), how many ran at once, and a strip chart of busy fraction over the burst.
Finding out where a pipeline idles; nothing here is on the product path."""
import argparse
import csv
import glob
import os
import sys


def load(dirname):
    rows = []  # (lane, start_ns, end_ns, bytes)
    for f in glob.glob(os.path.join(dirname, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0]
            name = name.replace("void ", "")
            rows.append((name[:44], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), 0))
    for f in glob.glob(os.path.join(dirname, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            d = r.get("Direction", "copy")
            d = {"MEMORY_COPY_HOST_TO_DEVICE": "copy host->device", "MEMORY_COPY_DEVICE_TO_HOST": "copy device->host",
                 "MEMORY_COPY_DEVICE_TO_DEVICE": "copy device->device"}.get(d, d)
            b = int(r.get("Bytes", r.get("Size", 0)) or 0) if ("Bytes" in r or "Size" in r) else 0
            rows.append((d, int(r["Start_Timestamp"]), int(r["End_Timestamp"]), b))
    return rows


def union(iv):
    iv = sorted(iv)
    busy, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        busy += cur_e - cur_s
    return busy


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--gap", type=float, default=30.0, help="idle gap (ms) that separates bursts")
    ap.add_argument("--bins", type=int, default=80)
    a = ap.parse_args()
    rows = load(a.dir)
    if not rows:
        sys.exit("no traces under " + a.dir)
    rows.sort(key=lambda r: r[1])
    # bursts
    bursts, cur, end = [], [rows[0]], rows[0][2]
    for r in rows[1:]:
        if r[1] - end > a.gap * 1e6:
            bursts.append(cur)
            cur = []
        cur.append(r)
        end = max(end, r[2])
    bursts.append(cur)
    print(f"{len(rows)} records, {len(bursts)} bursts (gap > {a.gap} ms); spans (ms): " +
          ", ".join(f"{(max(x[2] for x in b) - b[0][1]) / 1e6:.1f}" for b in bursts[-8:]))
    b = bursts[-1]
    t0, t1 = b[0][1], max(x[2] for x in b)
    span = t1 - t0
    print(f"last burst: {span / 1e6:.2f} ms, {len(b)} records")
    lanes = {}
    for name, s, e, nb in b:
        lanes.setdefault(name, []).append((s, e, nb))
    marks = " .:-=+*#%@"
    print(f"{'lane':46s} {'n':>6s} {'busy ms':>8s} {'busy %':>6s} {'sum ms':>8s} {'avg us':>8s} {'GB':>6s} {'GB/s busy':>9s}  strip ({a.bins} bins of {span / a.bins / 1e6:.2f} ms)")
    for name, iv in sorted(lanes.items(), key=lambda kv: -union([(s, e) for s, e, _ in kv[1]])):
        u = union([(s, e) for s, e, _ in iv])
        tot = sum(e - s for s, e, _ in iv)
        nbytes = sum(x[2] for x in iv)
        bins = [0.0] * a.bins
        w = span / a.bins
        for s, e, _ in iv:
            i0, i1 = int((s - t0) / w), min(a.bins - 1, int((e - t0) / w))
            for i in range(i0, i1 + 1):
                lo, hi = t0 + i * w, t0 + (i + 1) * w
                bins[i] += max(0, min(e, hi) - max(s, lo)) / w
        strip = "".join(marks[min(len(marks) - 1, int(round(min(x, 1.0) * (len(marks) - 1))))] for x in bins)
        gbs = f"{nbytes / u:9.1f}" if nbytes and u else " " * 9
        print(f"{name:46s} {len(iv):6d} {u / 1e6:8.2f} {100 * u / span:6.1f} {tot / 1e6:8.2f} {tot / len(iv) / 1e3:8.1f} {nbytes / 1e9:6.2f} {gbs}  |{strip}|")
    allu = union([(s, e) for _, s, e, _ in b])
    print(f"anything at all busy: {allu / 1e6:.2f} ms = {100 * allu / span:.1f} % of the burst")


if __name__ == "__main__":
    main()
