cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r06d
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r06d/gpu_suite.txt; tail -5 gpurun_out/r06d/gpu_suite.txt
