#!/bin/bash
# (1) C2 in motion per kernel with the run-record scatter; (2) C3 in motion: medium tile (5 workgroups per CU, 1032 slots) vs heavy class (4 per CU, bigger tile)
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD
export TMPDIR=/tmp
PMC_SETS=0 tools/prof.sh r05_c2_motion --presteps 2500 > /dev/null 2>&1
head -14 gpurun_out/prof_r05_c2_motion/summary.txt > gpurun_out/r05_c2_motion_trace.txt
rm -rf gpurun_out/prof_r05_c2_motion/trace
for v in base nomedium; do
  if [ $v = nomedium ]; then export SPH_HIP_LIB=$R/sph_project_amd/variants/libsph_hip_nomedium.so; else unset SPH_HIP_LIB; fi
  python bench.py --config c3 --presteps 1000 --steps 20 --warmup 3 --no-cpu-baseline --no-extras --motion-step 0 --all-kernels > gpurun_out/r05_c3m_$v.json 2> gpurun_out/r05_c3m_$v.err
  python - gpurun_out/r05_c3m_$v.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("C3 2+2 fixed, step 1000+ (%s): %.4f ms/step, fallback blocks %s" % (sys.argv[2], d["ms_per_step"], d["config"].get("lds_fallback_blocks_last_step")))
PY
  grep "launches" gpurun_out/r05_c3m_$v.err | sed 's/^/    /'
done
cat gpurun_out/r05_c2_motion_trace.txt
