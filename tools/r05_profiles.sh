set -u
cd $GRAFT_REPO_ROOT
tools/prof.sh r05_c3 --config c3 > gpurun_out/prof_r05_c3.log 2>&1
tools/prof.sh r05_c3_motion --config c3 --presteps 1000 --measured-iterations > gpurun_out/prof_r05_c3_motion.log 2>&1
BENCH_CMD="python $GRAFT_REPO_ROOT/tools/bench_c5.py --no-events --steps 10 --warmup 2" tools/prof.sh r05_c5 > gpurun_out/prof_r05_c5.log 2>&1
# keep the merged output small: drop the raw CSVs, keep summaries
for d in gpurun_out/prof_r05_c3 gpurun_out/prof_r05_c3_motion gpurun_out/prof_r05_c5; do rm -rf $d/trace $d/pmc1 $d/pmc2 $d/pmc3 $d/pmc4; done
tail -5 gpurun_out/prof_r05_c3/summary.txt
