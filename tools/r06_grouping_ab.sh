#!/bin/bash
# A/B of the staging groups of the neighbour passes (round 6): outer runs mixed (SPH_RUN_GROUPING=1, the default of the fast build on unsharded
# grids with nz >= 40) against the x-offset groups (SPH_RUN_GROUPING=0)
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_hip_wcsph.py tests/test_hip_golden.py tests/test_big_golden.py tests/test_hip_solvers.py tests/test_hip_round2.py -m gpu -x -q -k "not million and not bench" 2>&1 | tail -3
BENCH_ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-extras --all-kernels" tools/ab.sh r06_grouping mixed="" xoff="SPH_RUN_GROUPING=0" mixed2="" xoff2="SPH_RUN_GROUPING=0" 2>&1 | grep -v "^    .*\(scan\|hash\|misc\|scatter\)"
for v in "" "SPH_RUN_GROUPING=0" "" "SPH_RUN_GROUPING=0"; do
  env $v X=1 python bench.py --config c3 --steps 50 --warmup 5 --no-cpu-baseline --no-extras --motion-step 1000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C3 2+2 [%-20s] %.4f ms/step from rest, %.4f at step 1000' % ('$v', d['ms_per_step'], d['in_motion']['ms_per_step']))"
done
for v in "" "SPH_RUN_GROUPING=0"; do
  env $v X=1 python tools/bench_c5.py --no-events --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C5 [%-20s] %.4f ms/step' % ('$v', d['ms_per_step']))"
  env $v X=1 python bench.py --config c4 --steps 50 --warmup 5 --no-cpu-baseline --no-extras --motion-step 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C4 [%-20s] %.4f ms/step' % ('$v', d['ms_per_step']))"
done
