#!/usr/bin/env python
"""Rebuild profiles/pmc_derived.json from a committed summary file (the text tools/prof_summary.py printed on the GPU box), for when the
large PMC CSVs did not travel back: python tools/pmc_from_summary.py profiles/<summary>.txt [--config c2]"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import prof_summary as ps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(path, config="c2"):
    txt = open(path).read()
    trace = {m.group(1): float(m.group(3)) * 1e3 for m in re.finditer(r"^(\S+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)$", txt, re.M)}
    block = txt.split("== PMC counters, average per launch ==")[1].split("== derived")[0]
    ctr, cur = {}, None
    for line in block.splitlines():
        if line and not line.startswith(" "):
            cur = line.strip(); ctr[cur] = {}
        elif line.strip():
            a, b = line.split(); ctr[cur][a] = float(b)
    out_path = os.path.join(ROOT, "profiles", "pmc_derived.json")
    doc = json.load(open(out_path)) if os.path.exists(out_path) else {}
    doc["_source"] = os.path.relpath(os.path.abspath(path), ROOT)
    doc.setdefault(config, {})
    for kernel, c in ctr.items():
        m = re.match(r"nbr_pass<(\w+)>", kernel)
        kid = ps.KERNEL_ID.get(m.group(1)) if m else None
        if not kid or kernel not in trace:
            continue
        d = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in ps.derive(trace[kernel], c).items() if v is not None}
        d["avg_us"] = round(trace[kernel] / 1e3, 3)
        doc[config][kid] = d
    json.dump(doc, open(out_path, "w"), indent=1, sort_keys=True)
    print(json.dumps(doc[config], indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[3] if len(sys.argv) > 3 and sys.argv[2] == "--config" else "c2")
