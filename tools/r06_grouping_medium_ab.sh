#!/bin/bash
# A/B: x-offset staging groups for the 20-28-byte solver walks only (SPH_RUN_GROUPING_MEDIUM=0), mixed groups for everything else (default)
cd ${GRAFT_REPO_ROOT:-.}
for v in "X=1" "SPH_RUN_GROUPING_MEDIUM=0" "X=1" "SPH_RUN_GROUPING_MEDIUM=0"; do
  env $v python bench.py --config c3 --steps 50 --warmup 5 --no-cpu-baseline --no-extras --motion-step 1000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C3 2+2 [%-28s] %.4f ms/step from rest, %.4f at step 1000' % ('$v', d['ms_per_step'], d['in_motion']['ms_per_step']))"
  env $v python bench.py --config c3 --measured-iterations --presteps 1000 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --motion-step 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C3 measured, step 1000+ [%-12s] %.4f ms/step' % ('$v', d['ms_per_step']))"
  env $v python tools/bench_c5.py --no-events --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C5 [%-28s] %.4f ms/step' % ('$v', d['ms_per_step']))"
  env $v python bench.py --method pcisph --steps 30 --warmup 5 --no-cpu-baseline --no-extras --motion-step 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C2 PCISPH 2 fixed [%-18s] %.4f ms/step' % ('$v', d['ms_per_step']))"
done
