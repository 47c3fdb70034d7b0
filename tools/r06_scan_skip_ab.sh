#!/bin/bash
# k_scan_final: empty tiles whose cells already hold the right constant are left alone (State::scan_tile_state) against SPH_SCAN_ALL_TILES=1
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r06_scan_skip
timeout 1200 python -m pytest tests/test_hip_wcsph.py tests/test_hip_golden.py tests/test_hip_round2.py tests/test_hip_rigid.py tests/test_hip_solvers.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r06_scan_skip/tests.txt
tools/ab.sh r06_scan_skip skip="" all="SPH_SCAN_ALL_TILES=1" skip2="" all2="SPH_SCAN_ALL_TILES=1" 2>&1 | tee gpurun_out/r06_scan_skip/summary.txt
