#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for w in 2 3 2 3; do
  SPH_CG_SPLIT_WAYS=$w python tools/bench_c5.py --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C5 split ways $w: %.4f ms/step, %.1f CG it/step, %.2f us per CG iteration (events)' % (d['ms_per_step'], d['cg_iterations_per_step'], 1e3*d['ms_per_cg_iteration']), {k:v for k,v in d['kernels_ms_per_step'].items() if k.startswith('cg')})"
done
python tools/bench_c5.py --no-events --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C5 default (no events): %.4f ms/step' % d['ms_per_step'])"
SPH_CG_SPLIT_WAYS=3 python tools/bench_c5.py --no-events --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C5 three ways (no events): %.4f ms/step' % d['ms_per_step'])"
python -m pytest tests -m gpu -x -q -k "implicit or visc or c5 or cg" 2>&1 | tail -3
python bench.py --presteps 2500 --steps 50 --warmup 5 --no-cpu-baseline --no-extras --all-kernels --motion-step 0 2>&1 | grep "launches\|ms_per_step" | cut -c1-200
