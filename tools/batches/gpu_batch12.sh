#!/bin/bash
O=gpurun_out/b12; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --durations=5 > $O/tests.log 2>&1; tail -30 $O/tests.log
SPH_COMM_TRANSPORT=shm timeout 600 python bench.py --gpus 2 --method pcisph --no-cpu-baseline --steps 10 --warmup 3 --repeats 1 --motion-step 0 > $O/bench_pcisph_2ranks.json 2> $O/bench_pcisph_2ranks.err; cut -c1-300 $O/bench_pcisph_2ranks.json; tail -3 $O/bench_pcisph_2ranks.err
