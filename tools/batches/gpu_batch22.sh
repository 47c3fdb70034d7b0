#!/bin/bash
O=gpurun_out/b22; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_round2.py -m gpu -q -k "bench" > $O/tests.log 2>&1; tail -4 $O/tests.log | cut -c1-300
python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 400 $O/bench_c2.json
bash tools/prof.sh r02d --motion-step 0 > $O/prof.log 2>&1; tail -5 $O/prof.log
