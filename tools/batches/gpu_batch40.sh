#!/bin/bash
O=gpurun_out/b40; mkdir -p $O
SPH_BENCH_FORCE_SLAB=1 SPH_COMM_TRANSPORT=rccl python bench.py --no-cpu-baseline --motion-step 0 --all-kernels 2>&1 >/dev/null | grep -v "No rigid\|NCCL\|^$" | head -20
python bench.py --no-cpu-baseline --motion-step 0 --all-kernels 2>&1 >/dev/null | grep -v "No rigid" | head -12
for v in 0 1; do SPH_BENCH_FORCE_SLAB=$v SPH_COMM_TRANSPORT=rccl python bench.py --no-cpu-baseline --motion-step 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    i=l.find('{\"metric')
    if i>=0: d=json.loads(l[i:]); print('force_slab=$v', d['ms_per_step'])"; done
