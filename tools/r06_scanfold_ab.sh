#!/bin/bash
# A/B of the scan fold (round 6): the tile sums of the prefix scan are left by whoever takes the histogram atomics (k_hash_count, the NextHash
# epilogue of the force pass, the slab kernels), so k_scan_reduce is not launched.  SPH_NO_SCAN_FOLD=1 = two scan launches per sort as before.
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_hip_wcsph.py tests/test_hip_golden.py tests/test_big_golden.py tests/test_hip_slab.py -m gpu -x -q 2>&1 | tail -3
BENCH_ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-extras" tools/ab.sh r06_scanfold new="" old="SPH_NO_SCAN_FOLD=1" new2="" old2="SPH_NO_SCAN_FOLD=1"
for v in "" "SPH_NO_SCAN_FOLD=1" "" "SPH_NO_SCAN_FOLD=1"; do
  env $v X=1 python bench.py --config c3 --steps 50 --warmup 5 --no-cpu-baseline --no-extras --motion-step 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C3 2+2 [$v] %.4f ms/step' % d['ms_per_step'])"
done
