cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r06d_long
(timeout 600 python tools/long_run_sort_ab.py 20000 2000 wcsph; timeout 600 python tools/long_run_sort_ab.py 6000 2000 dfsph; timeout 600 python tools/long_run_sort_ab.py 6000 2000 pcisph) 2>&1 | grep -v "No rigid" | tee gpurun_out/r06d_long/sort_ab.txt
(timeout 300 python tools/long_run.py 20000 wcsph; timeout 300 python tools/long_run.py 6000 dfsph; timeout 300 python tools/long_run.py 6000 pcisph) 2>&1 | grep -v "No rigid" | tee gpurun_out/r06d_long/long_run.txt
