#!/bin/bash
O=gpurun_out/b16; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --durations=5 > $O/tests.log 2>&1; tail -12 $O/tests.log | cut -c1-300
SPH_COMM_TRANSPORT=shm python tools/bench_c5.py --help 2>&1 | head -20
