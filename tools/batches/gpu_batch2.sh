#!/bin/bash
# GPU batch 2 (round 2): new GPU tests, packed SoA phase-1 microbenchmark, the new bench line
O=gpurun_out/b2; mkdir -p $O
( cd tools/mb && timeout 300 ./mb_issue_noslp > ../../$O/mb_issue_noslp.txt 2>&1 )
timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > $O/tests.log 2>&1; tail -40 $O/tests.log
timeout 600 python bench.py --all-kernels > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json
