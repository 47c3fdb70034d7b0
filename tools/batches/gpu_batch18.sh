#!/bin/bash
O=gpurun_out/b18; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_wcsph.py tests/test_hip_solvers.py tests/test_hip_golden.py -m gpu -q -x > $O/tests.log 2>&1; tail -3 $O/tests.log
python bench.py --no-cpu-baseline > $O/c2.json 2> $O/c2.err
python bench.py --no-cpu-baseline --config c4 > $O/c4.json 2> $O/c4.err
python bench.py --no-cpu-baseline --config c3 > $O/c3.json 2> $O/c3.err
for f in c2 c4 c3; do python - <<PY
import json
d=json.loads(open("$O/$f.json").read().strip().splitlines()[-1])
m=d.get("in_motion") or {}
print("$f", round(d["ms_per_step"],4), d["config"].get("lds_fallback_blocks_last_step"), "motion", m.get("ms_per_step"), m.get("lds_fallback_blocks_last_step"))
PY
done
