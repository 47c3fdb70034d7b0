#!/bin/bash
O=gpurun_out/b39; mkdir -p $O
python bench.py --no-cpu-baseline --config c3 > $O/bench_c3.json 2>/dev/null
python bench.py --no-cpu-baseline --method pcisph > $O/bench_pcisph.json 2>/dev/null
for f in c3 pcisph; do python - <<PY
import json
d=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1]); m=d.get("in_motion") or {}
print("$f", round(d["ms_per_step"],4), m.get("ms_per_step"), m.get("lds_fallback_blocks_last_step"))
PY
done
