# one-rank slab overhead, transports side by side, 2 ranks on one GPU, kernel trace of the one-rank slab step
T=${1:-r03d}
mkdir -p gpurun_out/$T; O=gpurun_out/$T
A="--steps 100 --warmup 10 --no-cpu-baseline --no-extras --motion-step 0"
python bench.py $A > $O/plain.json 2> $O/plain.err
SPH_BENCH_FORCE_SLAB=1 python bench.py $A > $O/slab1_push.json 2> $O/slab1_push.err
SPH_BENCH_FORCE_SLAB=1 SPH_SLAB_ASYNC=0 python bench.py $A > $O/slab1_push_sync.json 2> $O/slab1_push_sync.err
SPH_BENCH_FORCE_SLAB=1 SPH_COMM_TRANSPORT=rccl python bench.py $A > $O/slab1_rccl.json 2> $O/slab1_rccl.err
SPH_COMM_TRANSPORT=shm+ipc python bench.py --gpus 2 $A > $O/two_push.json 2> $O/two_push.err
SPH_COMM_TRANSPORT=shm+ipc SPH_SLAB_ASYNC=0 python bench.py --gpus 2 $A > $O/two_push_sync.json 2> $O/two_push_sync.err
SPH_COMM_TRANSPORT=shm python bench.py --gpus 2 $A > $O/two_shm.json 2> $O/two_shm.err
R=$(pwd)
cd /tmp; export TMPDIR=/tmp
SPH_BENCH_FORCE_SLAB=1 SPH_BENCH_NO_EVENTS=1 timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/trace_slab1 -o t --output-format csv -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --motion-step 0 > $R/$O/trace.log 2>&1
cd $R
python - $O <<'PY'
import json,glob,sys,csv
O=sys.argv[1]
for f in sorted(glob.glob(O+'/*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1].ljust(24), '%.4f ms/step' % d['ms_per_step'], d['config']['parallelism'])
    except Exception as e: print(f, 'ERR', e)
f=glob.glob(O+'/trace_slab1/**/*kernel_stats.csv',recursive=True)
if f:
    for r in list(csv.DictReader(open(f[0])))[:22]:
        print(r['Name'].replace('sph_fast_ns::','')[:64].ljust(64), r['Calls'].rjust(5), '%9.1f us' % (float(r['AverageNs'])/1e3), r['Percentage'])
PY
