#!/bin/bash
O=gpurun_out/b24; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -k "implicit or c5" > $O/tests.log 2>&1; tail -4 $O/tests.log | cut -c1-300
for rep in 1 2; do
SPH_NO_CG_SPLIT=1 python tools/bench_c5.py --no-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nosplit', d['ms_per_step'], d['cg_iterations_per_step'])"
python tools/bench_c5.py --no-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('split  ', d['ms_per_step'], d['cg_iterations_per_step'])"
done
python tools/bench_c5.py 2>/dev/null | tail -1 > $O/c5_per_kernel.json; python -c "
import json; d=json.loads(open('$O/c5_per_kernel.json').read()); print(d['ms_per_step'], d['kernels_ms_per_step'])"
