mkdir -p gpurun_out/r04f
for o in xyz zxy; do echo "== order $o"; SPH_AXIS_ORDER=$o python tools/slab_size_probe.py --steps 200 2>/dev/null | tee gpurun_out/r04f/probe_$o.json; done
cd /tmp && export TMPDIR=/tmp
for o in xyz zxy; do
SPH_AXIS_ORDER=$o rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04f/trace_$o -- python $GRAFT_REPO_ROOT/tools/slab_size_probe.py --steps 200 > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/r04f/trace_$o -name "*kernel_stats.csv" | head -1); echo "== kernel stats $o ($f)"; head -14 $f | cut -c1-140
find $GRAFT_REPO_ROOT/gpurun_out/r04f/trace_$o -name "*.csv" -size +2M -delete; find $GRAFT_REPO_ROOT/gpurun_out/r04f/trace_$o -name "*.db" -delete
done
