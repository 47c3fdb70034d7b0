#!/bin/bash
O=gpurun_out/b20; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --durations=5 > $O/tests.log 2>&1; tail -10 $O/tests.log | cut -c1-300
python bench.py --no-cpu-baseline --config c4 > $O/c4.json 2> $O/c4.err
python tools/bench_c5.py --no-events > $O/c5.json 2> $O/c5.err; tail -c 600 $O/c5.json
python - <<PY
import json
d=json.loads(open("$O/c4.json").read().strip().splitlines()[-1]); m=d.get("in_motion") or {}
print("c4", d["ms_per_step"], d["config"].get("lds_fallback_blocks_last_step"), m.get("ms_per_step"), m.get("lds_fallback_blocks_last_step"))
PY
