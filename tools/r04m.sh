O=gpurun_out/r04m; mkdir -p $O
timeout 300 python -m pytest tests/test_hip_slab.py -x -q -k "cut_that_moves or slow_axis" > $O/new_slab_tests.txt 2>&1; tail -3 $O/new_slab_tests.txt; grep -h "big-layer" $O/new_slab_tests.txt
timeout 200 python -m pytest tests/test_hip_golden.py -x -q > $O/golden.txt 2>&1; tail -2 $O/golden.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver_line.json 2> $O/bench_driver_line.err; tail -c 300 $O/bench_driver_line.json
bash tools/prof.sh r04 > $O/prof.log 2>&1; tail -5 $O/prof.log
mkdir -p $O/prof; cp gpurun_out/prof_r04/summary.txt $O/prof/ 2>/dev/null; find gpurun_out/prof_r04 -name "*.csv" -size +1M -delete; find gpurun_out/prof_r04 -name "*.db" -delete; du -sh gpurun_out/prof_r04
