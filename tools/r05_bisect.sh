#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
T='tests/test_hip_slab.py::test_c4_sharded_over_8_ranks_matches_oracle[shm+ipc-jitter_vz]'
for v in "X=1" "SPH_NO_SLAB_FUSED_FIELDS=1" "SPH_NO_SLAB_PRESEND=1" "SPH_SLAB_ASYNC=0"; do
  echo "== $v"; env $v python -m pytest "$T" -m gpu -x -q -s 2>&1 | grep -E "passed|failed|8 ranks on one|assert \(" | cut -c1-400
done
