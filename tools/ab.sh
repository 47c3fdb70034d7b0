#!/bin/bash
# One parametrised A/B recipe for the GPU box (replaces the 65 one-off tools/batches/*.sh of rounds 1-3).
#   tools/ab.sh <out-dir-under-gpurun_out> <label>=<env assignments or ""> ...   [BENCH_ARGS="..."]
# Every variant is one run of bench.py (same box, back to back) with its environment switches ("Switches"); prints
# ms/step from rest and in motion and the per-kernel table of the two neighbour walks.
#   tools/ab.sh r04a base="" noperm="SPH_NO_LANE_PERM=1" variant="SPH_HIP_LIB=sph_project_amd/variants/libsph_hip_x.so"
set -u
O=gpurun_out/$1; shift; mkdir -p "$O"
A=${BENCH_ARGS:---steps 100 --warmup 10 --no-cpu-baseline --no-extras --all-kernels}
for spec in "$@"; do
  label=${spec%%=*}; envs=${spec#*=}
  env $envs python bench.py $A > "$O/$label.json" 2> "$O/$label.err"
  python - "$O/$label.json" "$label" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
m = d.get("in_motion") or {}
print("%-16s %.4f ms/step from rest  %s in motion" % (sys.argv[2], d["ms_per_step"], ("%.4f" % m["ms_per_step"]) if m else "-"))
PY
  grep -h "density\|wcsph_forces\|scatter\|scan\|hash\|misc" "$O/$label.err" | sed 's/^/    /'
done
