#!/usr/bin/env python
"""Condense rocprofv3 CSV output (tools/prof.sh) into a per-kernel table: launches, avg/total
duration from the kernel trace, and PMC counter averages per launch; then the figures derived from them
(what bounds a kernel: the formulas are printed with the numbers so that a reader can redo them).

    python tools/prof_summary.py gpurun_out/prof_<tag> [--json profiles/pmc_derived.json --config c2 --source profiles/<summary>.txt]

--json merges {config: {kernel id: {...}}} into the given file: HBM bytes per launch and the secondary-bound figures
bench.py attaches to its `roofline` object (labelled there as "from profiles/, not this run")."""
import json
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


# chip constants (MI355X_MICROARCH.md): 256 CUs, 4 SIMDs each, 8 XCDs; a wave64 VALU instruction occupies its SIMD for 2 cycles
N_CU, N_SIMD, N_XCD, VALU_CYC = 256, 1024, 8, 2.0
KERNEL_ID = {"DensityPass": "density", "WcsphForcePass": "wcsph_forces", "NonPressurePass": "non_pressure", "PressurePass": "pressure_integrate",
             "DfsphDensityAlphaPass": "dfsph_density_alpha", "DfsphRhoAdvPass": "dfsph_rho_adv", "DfsphCorrectPass": "dfsph_correct",
             "PcisphRhoStarPass": "pcisph_rho_star", "PcisphPressureAccelPass": "pcisph_pressure_accel", "CgApPass": "cg_ap"}


def derive(avg_ns, c):
    """c: counter averages per launch.  Returns the derived figures (None where a counter is missing)."""
    d = {}
    g = c.get("GRBM_GUI_ACTIVE")
    cyc = g / N_XCD if g else None                       # GRBM_GUI_ACTIVE is summed over the 8 XCDs
    d["kernel_cycles"] = cyc
    d["eff_clock_ghz"] = cyc / avg_ns if cyc else None   # cycles per ns
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:          # KiB; gfx950: FETCH_SIZE counts 64 B per 128 B request -> x2 (guide)
        d["hbm_bytes_per_launch"] = int((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
    if cyc and "SQ_INSTS_VALU" in c:
        d["valu_issue_frac"] = c["SQ_INSTS_VALU"] / N_SIMD * VALU_CYC / cyc
    if c.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in c:
        d["waves_parked_frac"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
    if cyc and "SQ_LDS_IDX_ACTIVE" in c:
        d["lds_active_frac"] = c["SQ_LDS_IDX_ACTIVE"] / (N_CU * cyc)
    if c.get("SQ_LDS_IDX_ACTIVE") and "SQ_LDS_BANK_CONFLICT" in c:
        d["lds_conflict_frac"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    return d


def main(out, json_path=None, config="c2", source=None):
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
        print("\n== derived (per launch; chip: 256 CUs, 1024 SIMDs, 8 XCDs) ==")
        print("   kernel_cycles = GRBM_GUI_ACTIVE / 8 XCDs; eff_clock = kernel_cycles / avg duration; hbm_bytes = (2 FETCH_SIZE + WRITE_SIZE) KiB")
        print("   valu_issue = SQ_INSTS_VALU / 1024 SIMDs x 2 cycles / kernel_cycles; waves_parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES")
        print("   lds_active = SQ_LDS_IDX_ACTIVE / (256 CUs x kernel_cycles); lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE")
        print("   (GRBM_GUI_ACTIVE also counts the ramp around a launch: the clock and the fractions are meaningful for kernels of ~100 us, not for 5-25 us ones)")
        derived = {}
        for k in sorted(ctr, key=lambda k: -rows.get(k, {"ns": 0})["ns"]):
            if k not in rows or not rows[k]["n"]:
                continue
            avg_ns = rows[k]["ns"] / rows[k]["n"]
            d = derive(avg_ns, {c: ctr[k][c] / cnt[k][c] for c in ctr[k]})
            d["avg_us"] = avg_ns / 1e3
            derived[k] = d
            print(f"{k:40s} " + "  ".join(f"{kk}={vv:.3f}" if isinstance(vv, float) else f"{kk}={vv}" for kk, vv in d.items() if vv is not None))
        if json_path:
            try:
                doc = json.load(open(json_path))
            except (OSError, ValueError):
                doc = {}
            ent = doc.setdefault(config, {})
            for k, d in derived.items():
                m = re.match(r"nbr_pass<(\w+)>", k)
                kid = KERNEL_ID.get(m.group(1)) if m else None
                if kid:
                    ent[kid] = {kk: vv for kk, vv in d.items() if vv is not None}
            doc["_source"] = source or out
            ent["_source"] = source or out   # (per configuration: the passes of C2, C3 and C5 are separate runs with separate summary files)
            doc["_note"] = ("per launch, from separate rocprofv3 --pmc passes of `bench.py` (tools/prof.sh), condensed by tools/prof_summary.py; "
                            "formulas in the summary file named by _source")
            json.dump(doc, open(json_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--json")
    ap.add_argument("--config", default="c2")
    ap.add_argument("--source")
    a = ap.parse_args()
    main(a.out, a.json, a.config, a.source)
