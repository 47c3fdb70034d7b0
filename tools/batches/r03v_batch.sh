O=gpurun_out/r03v; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
A="--steps 200 --warmup 20 --no-cpu-baseline --no-extras --motion-step 0"
for m in 0 5; do
SPH_DEBUG_MODE=$m python bench.py $A --config c1 > $O/c1_mode$m.json 2>/dev/null
SPH_DEBUG_MODE=$m SPH_COMM_TRANSPORT=shm+ipc python bench.py --gpus 4 --scaling strong --no-c4 --steps 100 --warmup 10 --no-cpu-baseline --no-extras --motion-step 0 > $O/c2_strong4_mode$m.json 2>/dev/null
done
python - $O <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/*.json')):
    d=json.load(open(f)); print(f.split('/')[-1].ljust(26), '%.4f ms/step' % d['ms_per_step'], d['config']['parallelism'], d['config']['pair_interactions_per_step'])
PY
