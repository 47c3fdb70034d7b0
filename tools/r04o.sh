O=gpurun_out/r04o; mkdir -p $O
timeout 300 python -m pytest tests/test_hip_slab.py -x -q -k "cut_that_moves or cuts_follow" > $O/rebalance_test.txt 2>&1; tail -3 $O/rebalance_test.txt; grep -h "big-layer\|rebalance:" $O/rebalance_test.txt | head
echo "== C2 in motion (steps 2500..), per kernel"; timeout 200 python bench.py --presteps 2500 --steps 100 --warmup 10 --no-cpu-baseline --no-extras --all-kernels > $O/motion.json 2> $O/motion.err; grep -v "No rigid" $O/motion.err | head -8
python -c "
import json; d=json.loads(open('$O/motion.json').read().strip().split('\n')[-1]); c=d['config']; print('in motion', d['ms_per_step'], 'neighbours', c['neighbours_per_particle'], 'fallback', c['lds_fallback_blocks_last_step'])"
export SPH_COMM_TRANSPORT=shm+ipc
cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/two_rank_trace -o t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --gpus 2 --steps 50 --warmup 10 --no-extras --motion-step 0 > /dev/null 2> $GRAFT_REPO_ROOT/$O/two_rank_trace.err; cd $GRAFT_REPO_ROOT
for f in $(find $O/two_rank_trace -name "*kernel_stats.csv" | head -2); do echo "== $f"; head -18 $f | cut -c1-160; done
find $O/two_rank_trace -name "*.csv" -size +512k -delete; du -sh $O
