#!/bin/bash
# Round 6 A/B at C3 (1.23 M particles, DFSPH, 2+2 fixed iterations): the position update hashes for the sort that follows (SPH_NO_NEXT_HASH=1: a
# k_hash_count launch per step as before) and fixed-iteration solves do not launch the residual's last reduction (SPH_FIXED_KEEP_RESIDUAL=1: as before)
cd ${GRAFT_REPO_ROOT:-.}
for v in "" "SPH_NO_NEXT_HASH=1" "SPH_FIXED_KEEP_RESIDUAL=1" "SPH_NO_NEXT_HASH=1 SPH_FIXED_KEEP_RESIDUAL=1 SPH_NO_SCAN_FOLD=1" "" "SPH_NO_NEXT_HASH=1" "SPH_FIXED_KEEP_RESIDUAL=1" "SPH_NO_NEXT_HASH=1 SPH_FIXED_KEEP_RESIDUAL=1 SPH_NO_SCAN_FOLD=1"; do
  env $v X=1 python bench.py --config c3 --steps 50 --warmup 5 --no-cpu-baseline --no-extras --motion-step 1000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C3 2+2 [%-70s] %.4f ms/step from rest, %.4f at step 1000' % ('$v', d['ms_per_step'], d['in_motion']['ms_per_step']))"
done
