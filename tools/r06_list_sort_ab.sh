#!/bin/bash
# (the three-way form: needs tools/experiments/r06_gather_rank_prep.patch applied -- in the tree "lists" IS rank kernel + plain gather and SPH_SORT_RANK_KERNEL does nothing)
# deterministic sort by run lists against the sort by run records (SPH_NO_RUN_LISTS=1), and the two forms of the list sort:
#   lists   = k_scan_final + k_gather_rank_prep (2 launches; the gathering workgroup finds its slots' cells and source particles itself)
#   rankk   = SPH_SORT_RANK_KERNEL=1: k_scan_final + k_sort_rank + k_gather_prep (3 launches, inverse map in between)
#   records = k_scan_final + k_scatter_index + k_scatter<true> + k_block_prep (4 launches: round 5)
# tests first, then a same-box A/B
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r06_list_sort
timeout 900 python -m pytest tests/test_hip_wcsph.py tests/test_hip_golden.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r06_list_sort/tests.txt
SPH_SORT_RANK_KERNEL=1 timeout 900 python -m pytest tests/test_hip_wcsph.py -m gpu -x -q -k "list_sort or next_hash or c2_full_size_20" 2>&1 | tail -2 | tee -a gpurun_out/r06_list_sort/tests.txt
tools/ab.sh r06_list_sort lists="" rankk="SPH_SORT_RANK_KERNEL=1" records="SPH_NO_RUN_LISTS=1" lists2="" rankk2="SPH_SORT_RANK_KERNEL=1" records2="SPH_NO_RUN_LISTS=1" 2>&1 | tee gpurun_out/r06_list_sort/summary.txt
for v in "X=1" "SPH_SORT_RANK_KERNEL=1" "SPH_NO_RUN_LISTS=1" "X=1" "SPH_SORT_RANK_KERNEL=1" "SPH_NO_RUN_LISTS=1"; do
  env $v python bench.py --config c3 --steps 50 --warmup 5 --no-cpu-baseline --no-extras --motion-step 1000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C3 2+2 [%-24s] %.4f ms/step from rest, %.4f at step 1000' % ('$v', d['ms_per_step'], d['in_motion']['ms_per_step']))"
done 2>&1 | tee -a gpurun_out/r06_list_sort/summary.txt
