#!/bin/bash
# deterministic sort by run lists (k_sort_rank + k_gather_prep) against the sort by run records (SPH_NO_RUN_LISTS=1): tests first, then a same-box A/B
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r06_list_sort
timeout 900 python -m pytest tests/test_hip_wcsph.py tests/test_hip_golden.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r06_list_sort/tests.txt
tools/ab.sh r06_list_sort lists="" records="SPH_NO_RUN_LISTS=1" lists2="" records2="SPH_NO_RUN_LISTS=1" 2>&1 | tee gpurun_out/r06_list_sort/summary.txt
for v in "X=1" "SPH_NO_RUN_LISTS=1" "X=1" "SPH_NO_RUN_LISTS=1"; do
  env $v python bench.py --config c3 --steps 50 --warmup 5 --no-cpu-baseline --no-extras --motion-step 1000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C3 2+2 [%-20s] %.4f ms/step from rest, %.4f at step 1000' % ('$v', d['ms_per_step'], d['in_motion']['ms_per_step']))"
done 2>&1 | tee -a gpurun_out/r06_list_sort/summary.txt
