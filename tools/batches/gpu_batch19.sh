#!/bin/bash
# same-box A/B of library variants: SPH_HIP_LIB selects the .so
O=gpurun_out/b19; mkdir -p $O
for rep in 1 2; do for v in base relaxocc nopost; do
  SPH_HIP_LIB=$PWD/sph_project_amd/variants/libsph_hip_$v.so python bench.py --no-cpu-baseline > $O/c2_$v$rep.json 2>/dev/null
  SPH_HIP_LIB=$PWD/sph_project_amd/variants/libsph_hip_$v.so python bench.py --no-cpu-baseline --config c3 > $O/c3_$v$rep.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$O/c2_$v$rep.json").read().strip().splitlines()[-1]); e=json.loads(open("$O/c3_$v$rep.json").read().strip().splitlines()[-1])
print("$v $rep  c2 rest %.4f motion %.4f (fallback %s)   c3 %.4f" % (d["ms_per_step"], d["in_motion"]["ms_per_step"], d["in_motion"]["lds_fallback_blocks_last_step"], e["ms_per_step"]))
PY
done; done
