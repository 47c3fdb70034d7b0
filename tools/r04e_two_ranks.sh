mkdir -p gpurun_out/r04e
python -m pytest tests/test_hip_slab.py tests/test_hip_rigid.py -x -q > gpurun_out/r04e/slab_suite.txt 2>&1; tail -5 gpurun_out/r04e/slab_suite.txt
export SPH_COMM_TRANSPORT=shm+ipc
A="--gpus 2 --steps 100 --warmup 10 --no-extras --motion-step 0"
for v in "ovl:" "nopresend:SPH_NO_SLAB_PRESEND=1" "noovl:SPH_NO_SLAB_OVERLAP=1"; do
  l=${v%%:*}; e=${v#*:}
  env $e python bench.py $A > gpurun_out/r04e/two_$l.json 2> gpurun_out/r04e/two_$l.err
  python -c "
import json,sys; d=json.loads(open('gpurun_out/r04e/two_$l.json').read().strip().split('\n')[-1]); print('two ranks one GPU', '$l', '%.4f ms/step' % d['ms_per_step'], d['config']['parallelism'])"
done
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04e/trace -- python $GRAFT_REPO_ROOT/bench.py --gpus 2 --steps 50 --warmup 10 --no-extras --motion-step 0 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r04e/trace.err; cd $GRAFT_REPO_ROOT
find gpurun_out/r04e/trace -name "*kernel_stats*" | head; for f in $(find gpurun_out/r04e/trace -name "*kernel_stats.csv" | head -2); do echo $f; head -16 $f | cut -c1-150; done
find gpurun_out/r04e/trace -name "*kernel_trace.csv" -size +100k | head -1 | xargs -I{} cp {} gpurun_out/r04e/kernel_trace_rank.csv; find gpurun_out/r04e/trace -name "*.csv" -size +3M -delete; du -sh gpurun_out/r04e
unset SPH_COMM_TRANSPORT
python tools/slab_size_probe.py --steps 200 2>/dev/null
