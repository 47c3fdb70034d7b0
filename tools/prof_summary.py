#!/usr/bin/env python
"""Condense rocprofv3 CSV output (tools/prof.sh) into a per-kernel table: launches, avg/total
duration from the kernel trace, and PMC counter averages per launch."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"k_nbr_pass<\s*(?:sph_\w+_ns::)?(\w+)", name)
    if m:
        return "nbr_pass<" + m.group(1) + ">"
    m = re.search(r"(k_\w+)", name)
    return m.group(1) if m else name[:40]


def main(out):
    rows = defaultdict(lambda: {"n": 0, "ns": 0})
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            rows[k]["n"] += 1
            rows[k]["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    total = sum(v["ns"] for v in rows.values()) or 1
    print("== kernel trace (rocprofv3 --kernel-trace --stats) ==")
    print(f"{'kernel':40s} {'launches':>9s} {'avg_us':>10s} {'total_ms':>10s} {'%':>6s}")
    for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["ns"]):
        print(f"{k:40s} {v['n']:9d} {v['ns'] / v['n'] / 1e3:10.1f} {v['ns'] / 1e6:10.3f} {100 * v['ns'] / total:6.1f}")
    ctr = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            ctr[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
    if ctr:
        print("\n== PMC counters, average per launch ==")
        names = sorted({c for k in ctr for c in ctr[k]})
        for k in sorted(ctr, key=lambda k: -rows.get(k, {"ns": 0})["ns"]):
            print(k)
            for c in names:
                if c in ctr[k]:
                    print(f"    {c:28s} {ctr[k][c] / cnt[k][c]:16.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
