cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r06d_c3
timeout 900 python -m pytest tests/test_hip_solvers.py tests/test_hip_golden.py tests/test_hip_round2.py -m gpu -x -q 2>&1 | tail -4
for v in "X=1" "SPH_SORT_MOVE_ALL=1" "X=1" "SPH_SORT_MOVE_ALL=1"; do
  env $v python bench.py --config c3 --steps 50 --warmup 5 --no-cpu-baseline --no-extras --motion-step 1000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C3 2+2 [%-24s] %.4f ms/step from rest, %.4f at step 1000' % ('$v', d['ms_per_step'], d['in_motion']['ms_per_step']))"
done 2>&1 | tee gpurun_out/r06d_c3/summary.txt
