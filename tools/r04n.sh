O=gpurun_out/r04n; mkdir -p $O
timeout 300 python -m pytest tests/test_hip_slab.py -x -q -k "cut_that_moves" > $O/rebalance_test.txt 2>&1; tail -3 $O/rebalance_test.txt; grep -h "big-layer\|---- rank\|SphError\|status" $O/rebalance_test.txt | head -20
rm -rf gpurun_out/prof_r04
bash tools/prof.sh r04 > $O/prof.log 2>&1; tail -3 $O/prof.log
find gpurun_out/prof_r04 -name "*.csv" -size +1M -delete; find gpurun_out/prof_r04 -name "*.db" -delete; du -sh gpurun_out/prof_r04
