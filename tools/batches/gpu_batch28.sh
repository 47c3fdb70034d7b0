#!/bin/bash
O=gpurun_out/b28; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -2
python bench.py --no-cpu-baseline --method pcisph --motion-step 0 > $O/pcisph.json 2>/dev/null
python bench.py --no-cpu-baseline --config c3 --measured-iterations --motion-step 0 > $O/c3_measured.json 2>/dev/null
python bench.py --no-cpu-baseline --method pcisph --measured-iterations --motion-step 0 > $O/pcisph_measured.json 2>/dev/null
for f in pcisph c3_measured pcisph_measured; do python - <<PY
import json
d=json.loads(open("$O/$f.json").read().strip().splitlines()[-1])
print("$f", round(d["ms_per_step"],4), d["config"].get("method"), {k:v for k,v in d["config"].items() if "iter" in k})
PY
done
