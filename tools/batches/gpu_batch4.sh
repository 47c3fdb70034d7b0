#!/bin/bash
# GPU batch 4 (round 2): SoA phase 1 -- correctness (full suite on the default build) and A/B of chunk size / occupancy
O=gpurun_out/b4; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q --durations=8 > $O/tests.log 2>&1; tail -30 $O/tests.log
for v in base q1w4 q1w5 q2w5; do
  if [ $v = base ]; then unset SPH_HIP_LIB; else export SPH_HIP_LIB=$PWD/sph_project_amd/variants/libsph_hip_$v.so; fi
  timeout 600 python bench.py --all-kernels --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err
  python - $O/bench_$v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]+".json").read().strip().splitlines()[-1])
print(sys.argv[1], "rest %.4f ms" % d["ms_per_step"], ["%.4f" % x for x in d["repeat_ms_per_step"]], "motion %.4f" % d["in_motion"]["ms_per_step"], "fallback", d["config"]["lds_fallback_blocks_last_step"], d["in_motion"]["lds_fallback_blocks_last_step"])
print(open(sys.argv[1]+".err").read()[:330])
PY
done
