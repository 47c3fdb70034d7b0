#!/bin/bash
# the sort's gather without the colours (kept at home, keyed by the particle id) and without the density the next pass recomputes, against SPH_SORT_MOVE_ALL=1
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r06_move_less
timeout 900 python -m pytest tests/test_hip_wcsph.py tests/test_hip_golden.py tests/test_hip_round2.py -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/r06_move_less/tests.txt
tools/ab.sh r06_move_less less="" all="SPH_SORT_MOVE_ALL=1" less2="" all2="SPH_SORT_MOVE_ALL=1" 2>&1 | tee gpurun_out/r06_move_less/summary.txt
