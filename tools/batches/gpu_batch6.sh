#!/bin/bash
# GPU batch 6 (round 2): DFSPH under slab sharding + the full suite; C5 without per-kernel events
O=gpurun_out/b6; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q --durations=6 > $O/tests.log 2>&1; tail -30 $O/tests.log
timeout 900 python tools/bench_c5.py --no-events > $O/bench_c5_noevents.json 2> $O/bench_c5.err; cut -c1-400 $O/bench_c5_noevents.json
SPH_COMM_TRANSPORT=shm timeout 600 python bench.py --gpus 2 --config c3 --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_c3_2ranks_shm.json 2> $O/bench_c3_2ranks.err; cut -c1-700 $O/bench_c3_2ranks_shm.json; tail -5 $O/bench_c3_2ranks.err
