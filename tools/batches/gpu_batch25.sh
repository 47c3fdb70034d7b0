#!/bin/bash
# timing experiment: dot-product phase 1 (variant "dot") vs HEAD ("noq"); per-kernel times at rest
O=gpurun_out/b25; mkdir -p $O
for rep in 1 2; do for v in noq dot; do
  SPH_HIP_LIB=$PWD/sph_project_amd/variants/libsph_hip_$v.so python bench.py --no-cpu-baseline --motion-step 0 > $O/c2_$v$rep.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$O/c2_$v$rep.json").read().strip().splitlines()[-1])
print("$v $rep  c2 rest %.4f  pairs/step %d  roofline kernel %s %.1f us" % (d["ms_per_step"], d["config"]["pair_interactions_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_us"]))
PY
done; done
for v in noq dot; do SPH_HIP_LIB=$PWD/sph_project_amd/variants/libsph_hip_$v.so python bench.py --no-cpu-baseline --motion-step 0 --all-kernels 2>&1 >/dev/null | grep -E "density|wcsph_forces" ; done
