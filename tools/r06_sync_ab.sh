#!/bin/bash
# A/B of the host's wait for the stream (round 6): publishing kernel + spin on pinned memory (default) against a bare hipStreamSynchronize
# (SPH_SLOW_SYNC=1) -- in the driver's configuration (3 x 20 timed steps between two waits each), at 100 steps, and for synchronous C5 steps
cd ${GRAFT_REPO_ROOT:-.}
for v in "X=1" "SPH_SLOW_SYNC=1" "X=1" "SPH_SLOW_SYNC=1"; do
  env $v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --motion-step 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C2 --steps 20  [%-16s] %.4f ms/step' % ('$v', d['ms_per_step']), [round(x,4) for x in d['repeat_ms_per_step']], 'forces %.1f us' % d['roofline']['avg_launch_us'])"
  env $v python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras --motion-step 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C2 --steps 100 [%-16s] %.4f ms/step' % ('$v', d['ms_per_step']), [round(x,4) for x in d['repeat_ms_per_step']])"
  env $v python tools/bench_c5.py --no-events --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C5             [%-16s] %.4f ms/step' % ('$v', d['ms_per_step']))"
done
env SPH_STEP_COPIES_STATS=1 python tools/bench_c5.py --no-events --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C5             [statistics copied at every step end, as until round 6] %.4f ms/step' % d['ms_per_step'])"
