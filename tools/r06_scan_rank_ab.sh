#!/bin/bash
# the rank of the sort by run lists done by the scan's threads (scan_rank_cells) against a kernel of its own (SPH_NO_SCAN_RANK=1: k_sort_rank)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r06_scan_rank
timeout 1200 python -m pytest tests/test_hip_wcsph.py tests/test_hip_golden.py tests/test_hip_round2.py tests/test_hip_rigid.py tests/test_hip_solvers.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r06_scan_rank/tests.txt
tools/ab.sh r06_scan_rank inscan="" kernel="SPH_NO_SCAN_RANK=1" inscan2="" kernel2="SPH_NO_SCAN_RANK=1" 2>&1 | tee gpurun_out/r06_scan_rank/summary.txt
for v in "X=1" "SPH_NO_SCAN_RANK=1"; do
  echo "slab_size_probe [$v]"; env $v timeout 200 python tools/slab_size_probe.py --steps 200 2>/dev/null | tail -3 | cut -c1-330
done 2>&1 | tee -a gpurun_out/r06_scan_rank/summary.txt
