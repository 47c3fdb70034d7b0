O=gpurun_out/r04g; mkdir -p $O
timeout 400 python -m pytest tests/test_hip_slab.py tests/test_hip_rigid.py -x -q > $O/slab_suite.txt 2>&1; tail -4 $O/slab_suite.txt
BENCH_ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-extras --all-kernels" timeout 300 tools/ab.sh r04g plain="" slab1="SPH_BENCH_FORCE_SLAB=1" slab1_nofuse="SPH_BENCH_FORCE_SLAB=1 SPH_NO_SLAB_PRESEND=1 SPH_NO_SLAB_FUSED_FIELDS=1"
export SPH_COMM_TRANSPORT=shm+ipc
A="--gpus 2 --steps 100 --warmup 10 --no-extras --motion-step 0"
for v in "fused:" "nopresend:SPH_NO_SLAB_PRESEND=1" "nofields:SPH_NO_SLAB_FUSED_FIELDS=1" "none:SPH_NO_SLAB_PRESEND=1 SPH_NO_SLAB_FUSED_FIELDS=1" "slow:SPH_SLAB_LAYOUT=slow"; do
  l=${v%%:*}; e=${v#*:}
  env $e timeout 120 python bench.py $A > $O/two_$l.json 2> $O/two_$l.err
  python -c "
import json,sys; d=json.loads(open('$O/two_$l.json').read().strip().split('\n')[-1]); print('two ranks one GPU', '$l', '%.4f ms/step' % d['ms_per_step'], d['config']['parallelism'])"
done
