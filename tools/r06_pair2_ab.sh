#!/bin/bash
# A/B of two accepted neighbours per merged-loop trip in the 20-28-byte solver walks (default since round 6) against one (-DSPH_P2_PAIR2_MEDIUM=0)
cd ${GRAFT_REPO_ROOT:-.}
V=sph_project_amd/variants/libsph_hip_nopair2.so
python -m pytest tests/test_hip_solvers.py tests/test_hip_golden.py tests/test_big_golden.py -m gpu -x -q 2>&1 | tail -2
for v in "X=1" "SPH_HIP_LIB=$V" "X=1" "SPH_HIP_LIB=$V"; do
  env $v python bench.py --config c3 --steps 50 --warmup 5 --no-cpu-baseline --no-extras --motion-step 1000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C3 2+2 [%-60s] %.4f ms/step from rest, %.4f at step 1000' % ('$v', d['ms_per_step'], d['in_motion']['ms_per_step']))"
  env $v python bench.py --config c3 --measured-iterations --presteps 1000 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --motion-step 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C3 measured, step 1000+ [%-44s] %.4f ms/step' % ('$v', d['ms_per_step']))"
  env $v python tools/bench_c5.py --no-events --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C5 [%-60s] %.4f ms/step' % ('$v', d['ms_per_step']))"
  env $v python bench.py --method pcisph --steps 30 --warmup 5 --no-cpu-baseline --no-extras --motion-step 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C2 PCISPH 2 fixed [%-50s] %.4f ms/step' % ('$v', d['ms_per_step']))"
done
