#!/usr/bin/env python
"""Basic-block census of one kernel of a device-only .s file (tools/isa_report.py --keep DIR writes them).

    python tools/isa_blocks.py /tmp/isa/fast.s 'DensityPassILb1ELb1EEELi1E' [--dump]

Lists every basic block of the kernel whose mangled name contains the pattern: label, line, VALU / SALU / LDS / VMEM counts, the
v_readlane / v_writelane (SGPR spill traffic) among the VALU, and where the block's terminating branch goes (a backward branch = loop).
Trip counts are not known statically: the census is combined by hand with the loop structure (profiles/r05_isa_census.txt)."""
import re
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    dump = "--dump" in sys.argv
    lines = open(path).read().splitlines()
    start = None
    for k, l in enumerate(lines):
        if re.match(r"^_Z\w+:", l) and pat in l.split(":")[0]:
            start = k
            break
    if start is None:
        sys.exit("kernel not found")
    blocks = []   # (label, first line, dict)
    cur = {"label": "entry", "line": start, "valu": 0, "lane": 0, "salu": 0, "ds": 0, "vmem": 0, "wait": 0, "trans": 0, "br": []}
    labels = {}
    for k in range(start + 1, len(lines)):
        s = lines[k].strip()
        if s.startswith(".Lfunc_end") or s.startswith("s_endpgm") and False:
            break
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            blocks.append(cur)
            cur = {"label": m.group(1), "line": k, "valu": 0, "lane": 0, "salu": 0, "ds": 0, "vmem": 0, "wait": 0, "trans": 0, "br": []}
            labels[m.group(1)] = len(blocks)
            continue
        if not s or s[0] in ";./":
            continue
        op = s.split()[0]
        if op.startswith("v_"):
            cur["valu"] += 1
            if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
                cur["lane"] += 1
            if re.match(r"v_(rcp|rsq|sqrt|exp|log)_", op):
                cur["trans"] += 1
        elif op.startswith("ds_"):
            cur["ds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            cur["vmem"] += 1
        elif op.startswith("s_waitcnt"):
            cur["wait"] += 1
        elif op.startswith(("s_cbranch", "s_branch")):
            cur["br"].append(s.split()[1])
            cur["salu"] += 1
        elif op.startswith("s_"):
            cur["salu"] += 1
    blocks.append(cur)
    order = {b["label"]: i for i, b in enumerate(blocks)}
    tot = sum(b["valu"] for b in blocks)
    print(f"{len(blocks)} blocks, {tot} VALU static ({sum(b['lane'] for b in blocks)} lane moves)")
    for i, b in enumerate(blocks):
        tgt = []
        for t in b["br"]:
            j = order.get(t, -1)
            tgt.append(f"{t}{'^' if 0 <= j <= i else ''}")
        print(f"{i:4d} {b['label']:12s} L{b['line'] - start:5d} valu {b['valu']:4d} (lane {b['lane']:3d}, trans {b['trans']:2d}) salu {b['salu']:4d} ds {b['ds']:3d} vmem {b['vmem']:3d} wait {b['wait']:2d}  -> {' '.join(tgt)}")
    if dump:
        for k in range(start, blocks[-1]["line"] + 400):
            if k < len(lines):
                print(f"{k - start:6d}: {lines[k]}")
                if lines[k].strip().startswith(".Lfunc_end"):
                    break


if __name__ == "__main__":
    main()
