#!/usr/bin/env python3
"""
Idle time between kernels from a rocprofv3 kernel trace (CSV): over the steady-state part of a
run, 1 - (sum of kernel durations) / (last end - first start), plus the distribution of the gaps.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py ...
    python tools/idle_from_trace.py gpurun_out/trace [--skip 0.5]

--skip: fraction of the trace (by kernel count) dropped from the front (model build, weight
packing, warm-up).  Decides whether capturing a sampler step in a hipGraph could buy anything
(SURVEY 8(f)3): it can only remove the gaps.
"""

import argparse
import csv
import glob
import os
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--skip", type=float, default=0.3)
    ap.add_argument("--pause-us", type=float, default=500.0)
    a = ap.parse_args()
    files = glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit("no *kernel_trace.csv under %s" % a.dir)
    rows = []
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    rows = rows[int(len(rows) * a.skip):]
    # steady state = the longest stretch without a host-side pause (a gap above --pause-us: volume
    # boundaries, synchronisations, the instrumented forward's event records)
    best, cur, end = (0, 0), 0, rows[0][1]
    for i, (s0, e0, _) in enumerate(rows):
        if s0 - end > a.pause_us * 1000:
            if i - cur > best[1] - best[0]:
                best = (cur, i)
            cur = i
        end = max(end, e0)
    if len(rows) - cur > best[1] - best[0]:
        best = (cur, len(rows))
    rows = rows[best[0]:best[1]]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    busy = 0
    gaps = []
    end = rows[0][0]
    for s, e, _ in rows:
        if s > end:
            gaps.append(s - end)
        busy += max(0, e - max(s, end))
        end = max(end, e)
    span = t1 - t0
    gaps.sort()
    n = len(gaps)
    print("%s: %d kernels over %.3f ms; busy %.3f ms; idle %.2f %%" % (
        os.path.basename(files[0]), len(rows), span / 1e6, busy / 1e6, 100.0 * (span - busy) / span))
    if n:
        print("gaps: %d, median %.2f us, p90 %.2f us, p99 %.2f us, max %.1f us, sum %.3f ms" % (
            n, gaps[n // 2] / 1e3, gaps[int(n * 0.9)] / 1e3, gaps[int(n * 0.99)] / 1e3, gaps[-1] / 1e3, sum(gaps) / 1e6))


if __name__ == "__main__":
    main()
