#!/bin/bash
# A/B of the solver loops' read-back (round 6): a one-wave kernel publishes residual + flags into pinned host memory and the host spins on the batch
# number (default) against hipMemcpyAsync D2H + hipStreamSynchronize (SPH_NO_LOOP_PUBLISH=1, as until round 5)
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_hip_solvers.py tests/test_hip_golden.py tests/test_big_golden.py tests/test_hip_round2.py -m gpu -x -q -k "not million and not bench and not wcsph" 2>&1 | tail -3
for v in "" "SPH_NO_LOOP_PUBLISH=1" "" "SPH_NO_LOOP_PUBLISH=1"; do
  env $v X=1 python tools/bench_c5.py --no-events --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C5 [%-24s] %.4f ms/step, %.1f CG iterations' % ('$v', d['ms_per_step'], d['cg_iterations_per_step']))"
  env $v X=1 python bench.py --config c3 --measured-iterations --steps 20 --warmup 5 --no-cpu-baseline --no-extras --motion-step 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C3 measured, from rest [%-24s] %.4f ms/step' % ('$v', d['ms_per_step']))"
  env $v X=1 python bench.py --config c3 --measured-iterations --presteps 1000 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --motion-step 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C3 measured, step 1000+ [%-24s] %.4f ms/step %s' % ('$v', d['ms_per_step'], d['config'].get('solver_iterations_last_step')))"
done
