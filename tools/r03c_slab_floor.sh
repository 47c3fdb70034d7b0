mkdir -p gpurun_out/r03c; O=gpurun_out/r03c
A="--steps 100 --warmup 10 --no-cpu-baseline --no-extras --motion-step 0"
python bench.py $A > $O/plain.json 2> $O/plain.err
SPH_BENCH_FORCE_SLAB=1 python bench.py $A > $O/slab1_push.json 2> $O/slab1_push.err
SPH_BENCH_FORCE_SLAB=1 SPH_SLAB_ASYNC=0 python bench.py $A > $O/slab1_push_sync.json 2> $O/slab1_push_sync.err
SPH_BENCH_FORCE_SLAB=1 SPH_COMM_TRANSPORT=rccl python bench.py $A > $O/slab1_rccl.json 2> $O/slab1_rccl.err
SPH_COMM_TRANSPORT=shm+ipc python bench.py --gpus 2 $A > $O/two_push.json 2> $O/two_push.err
SPH_COMM_TRANSPORT=shm+ipc SPH_SLAB_ASYNC=0 python bench.py --gpus 2 $A > $O/two_push_sync.json 2> $O/two_push_sync.err
SPH_COMM_TRANSPORT=shm python bench.py --gpus 2 $A > $O/two_shm.json 2> $O/two_shm.err
cd /tmp; export TMPDIR=/tmp
SPH_BENCH_FORCE_SLAB=1 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace_slab1 -o t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --motion-step 0 > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03c/*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], d['ms_per_step'], d['config']['parallelism'], d.get('c2_strong_scaling',{}).get('ms_per_step'))
    except Exception as e: print(f, 'ERR', e)
PY
python tools/prof_summary.py gpurun_out/r03c/trace_slab1 2>/dev/null | head -30 || true
ls gpurun_out/r03c/trace_slab1 | head
