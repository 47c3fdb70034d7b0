#!/usr/bin/env python3
"""Build-time gate: read hipcc's -Rpass-analysis=kernel-resource-usage remarks of one kernel
translation unit (the Makefile redirects them into build/*.resources) and fail if any kernel spills VGPRs or uses
scratch.  SGPR spills (v_writelane into a spare VGPR) are reported, not fatal.  Compiler diagnostics other than the
resource remarks are passed through to stderr.

    python3 tools/check_spills.py build/kernels_fast.resources
"""
import os
import re
import subprocess
import sys


def main(path):
    text = open(path, errors="replace").read()
    kernels, cur = {}, None
    other = []
    for line in text.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z /\[\]]+?):\s+(\S+) \[-Rpass-analysis", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
            continue
        if "[-Rpass-analysis=kernel-resource-usage]" in line or re.match(r"^\s*\d*\s*\|", line) or not line.strip():
            continue
        other.append(line)
    if other:
        sys.stderr.write("\n".join(other) + "\n")
    if any(re.search(r"\berror\b", l) for l in other) or not kernels:
        sys.exit(1)
    bad = {k: v for k, v in kernels.items() if int(v.get("VGPRs Spill", "0")) or int(v.get("ScratchSize [bytes/lane]", "0"))}
    sg = sum(1 for v in kernels.values() if int(v.get("SGPRs Spill", "0")))
    print(f"{path}: {len(kernels)} kernels, {len(bad)} with VGPR spills / scratch, {sg} with SGPR spills (to VGPR lanes)")
    if bad and os.environ.get("SPH_ALLOW_SPILLS"):
        print(f"  (SPH_ALLOW_SPILLS set: A/B build, not failing)")
        return
    if bad:
        try:
            names = subprocess.run(["c++filt"], input="\n".join(bad), text=True, capture_output=True).stdout.splitlines()
        except OSError:
            names = list(bad)
        for n, (k, v) in zip(names, bad.items()):
            print(f"  SPILL {n[:140]}: VGPRs Spill {v.get('VGPRs Spill')}, scratch {v.get('ScratchSize [bytes/lane]')} B/lane", file=sys.stderr)
        sys.exit(1)


if __name__ == "__main__":
    main(sys.argv[1])
