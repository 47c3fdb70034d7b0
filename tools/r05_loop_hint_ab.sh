#!/bin/bash
# A/B of the predictive first batch of device_loop (sph_steps.hpp): C5 (CG, ~41 iterations per step) and C3 with the reference's own stop
# tests in motion (~26 density iterations per step).  SPH_NO_LOOP_HINT=1 = batches 2, 4, 8, 8, ... as in round 4.
cd ${GRAFT_REPO_ROOT:-.}
for v in hint nohint hint2 nohint2; do
  case $v in nohint*) export SPH_NO_LOOP_HINT=1;; *) unset SPH_NO_LOOP_HINT;; esac
  python tools/bench_c5.py --no-events --steps 20 --warmup 3 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C5 $v', round(d['ms_per_step'],4), 'ms/step', d['cg_iterations_per_step'], 'CG iterations')"
  python bench.py --config c3 --measured-iterations --presteps 1000 --steps 20 --warmup 3 --no-cpu-baseline --no-extras --motion-step 0 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C3 in motion $v', round(d['ms_per_step'],4), 'ms/step', {k:v for k,v in d['config'].items() if 'iter' in k})"
done
