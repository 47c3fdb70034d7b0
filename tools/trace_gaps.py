#!/usr/bin/env python
"""Where the wall time of a step goes that is NOT kernel time: reads a rocprofv3 --kernel-trace CSV, takes the last `--tail-ms` of the
trace (the timed region of a bench run), and prints busy time, idle time, and the idle gaps grouped by (kernel before -> kernel after).

    rocprofv3 --kernel-trace -d DIR -o trace --output-format csv -- python bench.py ...
    python tools/trace_gaps.py DIR [--tail-ms 100]
"""
import argparse
import csv
import glob
import os
import re
from collections import defaultdict


def short(name):
    m = re.search(r"k_nbr_pass<\s*(?:sph_\w+_ns::)?(\w+)", name)
    if m:
        return "nbr<" + m.group(1) + ">"
    m = re.search(r"(k_\w+)", name)
    return m.group(1) if m else name[:32]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--tail-ms", type=float, default=100.0)
    ap.add_argument("--before", default="", help="ignore trailing kernels whose name contains this string")
    ap.add_argument("--max-gap-us", type=float, default=0.0, help="list only gaps shorter than this (drops host read-backs / set-up when the whole trace is analysed)")
    a = ap.parse_args()
    rows = []
    for f in glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    if not rows:
        print("no kernel trace under", a.dir)
        return
    if a.before:   # drop the trailing kernels whose name contains this (bench.py ends with a 1 GiB copy-rate measurement)
        while rows and a.before in rows[-1][2]:
            rows.pop()
    t_end = max(r[1] for r in rows)
    rows = [r for r in rows if r[0] >= t_end - a.tail_ms * 1e6]
    span = rows[-1][1] - rows[0][0]
    busy = 0
    gaps = defaultdict(lambda: [0, 0])
    cur_end = rows[0][0]
    prev = "(start)"
    for s, e, k in rows:
        if s > cur_end:
            if not a.max_gap_us or (s - cur_end) < a.max_gap_us * 1e3:
                g = gaps[(prev, k)]
                g[0] += 1; g[1] += s - cur_end
            busy += e - s
        else:
            busy += max(0, e - max(s, cur_end))
        if e > cur_end:
            cur_end = e
            prev = k
    idle = span - busy
    print(f"last {span / 1e6:.2f} ms of the trace: {len(rows)} kernels, busy {busy / 1e6:.3f} ms ({100 * busy / span:.1f} %), idle {idle / 1e6:.3f} ms")
    print(f"{'gap between':62s} {'count':>7s} {'avg_us':>8s} {'total_ms':>9s} {'% of span':>9s}")
    for (p, k), (n, ns) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:24]:
        print(f"{p + ' -> ' + k:62s} {n:7d} {ns / n / 1e3:8.2f} {ns / 1e6:9.3f} {100 * ns / span:9.2f}")


if __name__ == "__main__":
    main()
