O=gpurun_out/r04p; mkdir -p $O
timeout 500 python -m pytest tests/test_hip_slab.py -x -q > $O/slab_suite.txt 2>&1; tail -3 $O/slab_suite.txt
export SPH_COMM_TRANSPORT=shm+ipc
A="--gpus 2 --steps 100 --warmup 10 --no-extras --motion-step 0"
for v in "fused:" "none:SPH_NO_SLAB_PRESEND=1 SPH_NO_SLAB_FUSED_FIELDS=1"; do
  l=${v%%:*}; e=${v#*:}
  env $e timeout 120 python bench.py $A > $O/two_$l.json 2> $O/two_$l.err
  python -c "
import json,sys; d=json.loads(open('$O/two_$l.json').read().strip().split('\n')[-1]); print('two ranks one GPU', '$l', '%.4f ms/step' % d['ms_per_step'], d['config']['parallelism'])"
done
cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/two_rank_trace -o t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --gpus 2 --steps 50 --warmup 10 --no-extras --motion-step 0 > /dev/null 2> $GRAFT_REPO_ROOT/$O/two_rank_trace.err; cd $GRAFT_REPO_ROOT
find $O/two_rank_trace -name "*.csv" -size +512k -delete
python - <<'PY'
import csv,re,glob
for f in glob.glob('gpurun_out/r04p/two_rank_trace/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        m=re.search(r'(k_\w+)(<[^>]*>)?', r['Name'])
        print('%-40s calls %5s avg %9.1f us' % ((m.group(0) if m else r['Name'])[:40], r['Calls'], float(r['AverageNs'])/1e3))
PY
