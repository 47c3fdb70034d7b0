#!/bin/bash
# (the three-way form: needs tools/experiments/r06_gather_rank_prep.patch applied -- in the tree "lists" IS rank kernel + plain gather and SPH_SORT_RANK_KERNEL does nothing)
# the three sorts (tools/r06_list_sort_ab.sh) where launches count most: C5 (2.17 M particles, ~106 k of them fluid) and unsharded blocks the size of a rank's share
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r06_list_sort
for v in "X=1" "SPH_SORT_RANK_KERNEL=1" "SPH_NO_RUN_LISTS=1" "X=1" "SPH_SORT_RANK_KERNEL=1" "SPH_NO_RUN_LISTS=1"; do
  env $v timeout 200 python tools/bench_c5.py --no-events --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C5 [%-24s] %.4f ms/step' % ('$v', d['ms_per_step']))"
done 2>&1 | tee gpurun_out/r06_list_sort/small.txt
for v in "X=1" "SPH_SORT_RANK_KERNEL=1" "SPH_NO_RUN_LISTS=1"; do
  echo "slab_size_probe [$v]"; env $v timeout 200 python tools/slab_size_probe.py --steps 200 2>/dev/null | tail -3
done 2>&1 | tee -a gpurun_out/r06_list_sort/small.txt
