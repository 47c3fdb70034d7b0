#!/bin/bash
O=gpurun_out/b11; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --durations=5 > $O/tests.log 2>&1; tail -25 $O/tests.log
timeout 900 python tools/bench_c5.py --no-events > $O/bench_c5_noevents.json 2> $O/bench_c5.err; cut -c1-420 $O/bench_c5_noevents.json
timeout 900 python tools/bench_c5.py > $O/bench_c5_events.json 2>> $O/bench_c5.err; python -c "
import json; d=json.loads(open('gpurun_out/b11/bench_c5_events.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_cg_iteration'], d['kernels_ms_per_step'])"
