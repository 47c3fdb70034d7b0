# round 3, batch y: full GPU suite on the committed build, profile (kernel trace + PMC passes) and the driver's bench line
O=gpurun_out/r03y; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -4 $O/tests.log
bash tools/prof.sh r03y --motion-step 0 > $O/prof.log 2>&1
timeout 900 python bench.py > $O/bench_driver_line.json 2> $O/bench.err; echo "bench rc=$?"
tail -c 1500 $O/bench_driver_line.json
