#!/bin/bash
# A/B of NextHash (the WCSPH force pass hashes for the next step's sort): SPH_NO_NEXT_HASH=1 = a k_hash_count launch per step as before
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_hip_wcsph.py tests/test_hip_golden.py tests/test_big_golden.py -m gpu -x -q 2>&1 | tail -3
BENCH_ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-extras" tools/ab.sh r05_nh new="" old="SPH_NO_NEXT_HASH=1" new2="" old2="SPH_NO_NEXT_HASH=1"
for v in "" "SPH_NO_NEXT_HASH=1" "" "SPH_NO_NEXT_HASH=1"; do
  env $v X=1 python bench.py --config c4 --steps 50 --warmup 5 --no-cpu-baseline --no-extras --motion-step 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C4 [$v] %.4f ms/step' % d['ms_per_step'])"
done
