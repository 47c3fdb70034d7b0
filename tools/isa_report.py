#!/usr/bin/env python
"""Per-kernel resource / instruction report of the gfx950 code objects (device-only -S of sph_kernels.hip).

    python tools/isa_report.py [--build fast|strict|both] [--check] [--filter k_nbr_pass] [--out FILE] [-D...]

For every kernel: VGPRs (allocation granule 8 -> waves/SIMD), SGPRs, spills, scratch, LDS bytes, and a static
instruction census (VALU, packed f32, transcendental, LDS reads/writes, global loads/stores, SALU, branches,
s_waitcnt).  --check exits non-zero if any kernel spills or uses scratch: __graft_entry__.build() runs it so that
a spilling k_nbr_pass instantiation fails the build.
"""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sph_project_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), text=True,
                             capture_output=True, check=True).stdout.splitlines()
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def short(name):
    name = re.sub(r"sph_(fast|strict)_ns::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(Consts.*$", "", name)
    name = name.replace("k_nbr_pass<", "nbr<").replace("(bool)1", "1").replace("(bool)0", "0").replace("true", "1").replace("false", "0")
    return name


def compile_asm(build, extra, out):
    flags = {"fast": ["-DSPH_FAST=1", "-ffp-contract=fast"], "strict": ["-DSPH_FAST=0", "-ffp-contract=off"]}[build]
    # same code generation flags as sph_project_amd/csrc/Makefile (COMMON + KFLAGS)
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-fno-slp-vectorize", "--cuda-device-only", "-S"] + flags + extra + \
          [os.path.join(CSRC, "sph_kernels.hip"), "-o", out]
    subprocess.check_call(cmd)


CLASSES = [
    ("v_pk", re.compile(r"^v_pk_")),
    ("trans", re.compile(r"^v_(rcp|rsq|sqrt|exp|log|sin|cos)_")),
    ("valu", re.compile(r"^v_")),
    ("ds_read", re.compile(r"^ds_read")),
    ("ds_write", re.compile(r"^ds_write")),
    ("ds_other", re.compile(r"^ds_")),
    ("vmem_ld", re.compile(r"^(global|buffer|flat|scratch)_load")),
    ("vmem_st", re.compile(r"^(global|buffer|flat|scratch)_store")),
    ("vmem_atomic", re.compile(r"^(global|buffer|flat)_atomic")),
    ("waitcnt", re.compile(r"^s_waitcnt")),
    ("barrier", re.compile(r"^s_barrier")),
    ("branch", re.compile(r"^s_(c)?branch")),
    ("smem", re.compile(r"^s_(load|buffer_load)")),
    ("salu", re.compile(r"^s_")),
]


def parse(path):
    text = open(path).read()
    kernels = {}
    # instruction census per function body
    cur = None
    for line in text.splitlines():
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            cur = m.group(1)
            kernels.setdefault(cur, {"ins": {c: 0 for c, _ in CLASSES}})
            kernels[cur]["ins"]["total"] = 0
            continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end") or line.strip().startswith(".end_amdhsa_kernel"):
            cur = None
            continue
        s = line.strip()
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        for c, rx in CLASSES:
            if rx.match(op):
                kernels[cur]["ins"][c] += 1
                if c in ("v_pk", "trans"):
                    kernels[cur]["ins"]["valu"] += 1
                break
        kernels[cur]["ins"]["total"] += 1
    # metadata
    for blk in re.split(r"\n  - \.agpr_count:", text)[1:]:
        def g(key, cast=int):
            m = re.search(r"\." + key + r":\s*(\S+)", blk)
            return cast(m.group(1)) if m else None
        name = g("name", str)
        if name is None:
            continue
        k = kernels.setdefault(name, {"ins": {}})
        k.update(vgpr=g("vgpr_count"), sgpr=g("sgpr_count"), spill=g("vgpr_spill_count"), sspill=g("sgpr_spill_count"),
                 scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size"))
    return {k: v for k, v in kernels.items() if "vgpr" in v}


def waves(vgpr):
    alloc = (vgpr + 7) // 8 * 8
    return min(8, 512 // max(alloc, 8))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", default="both", choices=["fast", "strict", "both"])
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--filter", default="")
    ap.add_argument("--out", default=None)
    ap.add_argument("--keep", default=None, help="directory to keep the .s files in")
    args, extra = ap.parse_known_args()
    builds = ["fast", "strict"] if args.build == "both" else [args.build]
    tmp = args.keep or "/tmp/isa_report"
    os.makedirs(tmp, exist_ok=True)
    lines = []
    bad = []
    for b in builds:
        path = os.path.join(tmp, b + ".s")
        compile_asm(b, extra, path)
        ks = parse(path)
        dm = demangle(list(ks))
        lines.append(f"== {b} build ({' '.join(extra)}) ==")
        lines.append(f"{'kernel':52s} {'vgpr':>4s} {'w/S':>3s} {'sgpr+spl':>8s} {'vspl':>4s} {'scr':>4s} {'lds':>6s} | {'total':>6s} {'valu':>6s} {'v_pk':>5s} {'trans':>5s} {'dsrd':>5s} {'dswr':>5s} {'vld':>4s} {'vst':>4s} {'salu':>5s} {'br':>4s} {'wait':>4s}")
        for name in sorted(ks, key=lambda n: short(dm[n])):
            v = ks[name]
            sn = short(dm[name])
            if args.filter and args.filter not in sn:
                continue
            i = v["ins"]
            lines.append(f"{sn[:52]:52s} {v['vgpr']:4d} {waves(v['vgpr']):3d} {v['sgpr']:4d}+{v.get('sspill') or 0:<3d} {v['spill']:4d} {v['scratch']:4d} {v['lds']:6d} | "
                         f"{i.get('total', 0):6d} {i.get('valu', 0):6d} {i.get('v_pk', 0):5d} {i.get('trans', 0):5d} {i.get('ds_read', 0):5d} {i.get('ds_write', 0):5d} "
                         f"{i.get('vmem_ld', 0):4d} {i.get('vmem_st', 0):4d} {i.get('salu', 0):5d} {i.get('branch', 0):4d} {i.get('waitcnt', 0):4d}")
            if v["spill"] or v["scratch"]:
                bad.append(f"{b}: {sn}: vgpr_spill {v['spill']} sgpr_spill {v.get('sspill')} scratch {v['scratch']}")
    text = "\n".join(lines)
    if args.out:
        open(args.out, "w").write(text + "\n")
    else:
        print(text)
    if bad:
        print("SPILLS / SCRATCH:\n  " + "\n  ".join(bad), file=sys.stderr)
        if args.check:
            sys.exit(1)


if __name__ == "__main__":
    main()
