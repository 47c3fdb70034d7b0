#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests -m gpu -x -q -k "visc or implicit or c5 or golden or solvers or round2 or rigid" 2>&1 | tail -3
for k in 1 2; do python tools/bench_c5.py --no-events --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C5 (no events): %.4f ms/step %.1f CG it' % (d['ms_per_step'], d['cg_iterations_per_step']))"; done
python tools/bench_c5.py --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C5 with events: %.4f ms/step, %.2f us per CG iteration' % (d['ms_per_step'], 1e3*d['ms_per_cg_iteration']), d['kernels_ms_per_step'])"
