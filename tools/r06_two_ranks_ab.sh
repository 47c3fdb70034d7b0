cd ${GRAFT_REPO_ROOT:-.}
for v in "X=1" "SPH_NO_SCAN_FOLD=1" "X=1" "SPH_NO_SCAN_FOLD=1"; do
  env $v SPH_COMM_TRANSPORT=shm+ipc python bench.py --gpus 2 --scaling weak --steps 20 --warmup 5 --no-cpu-baseline --no-extras --motion-step 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('2 ranks x 1.23M weak [$v] %.4f ms/step' % d['ms_per_step'], [round(x,4) for x in d['repeat_ms_per_step']])"
done
for v in "X=1" "SPH_NO_SCAN_FOLD=1"; do
  env $v SPH_COMM_TRANSPORT=shm+ipc python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --motion-step 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('2 ranks strong C2 [$v] %.4f ms/step' % d['ms_per_step'], [round(x,4) for x in d['repeat_ms_per_step']])"
done
