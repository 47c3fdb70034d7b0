#!/bin/bash
# final numbers of round 2 from HEAD: test suite, bench lines, rocprofv3 profile
O=gpurun_out/b33; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; tail -3 $O/tests.log | cut -c1-200
python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
python bench.py --no-cpu-baseline --config c3 > $O/bench_c3.json 2>/dev/null
python bench.py --no-cpu-baseline --config c4 > $O/bench_c4.json 2>/dev/null
python tools/bench_c5.py --no-events 2>/dev/null | tail -1 > $O/bench_c5.json
python tools/bench_c5.py 2>/dev/null | tail -1 > $O/bench_c5_per_kernel.json
bash tools/prof.sh r02e --motion-step 0 > $O/prof.log 2>&1
for f in c2 c3 c4; do python - <<PY
import json
d=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1]); m=d.get("in_motion") or {}
print("$f", round(d["ms_per_step"],4), m.get("ms_per_step"), d["roofline"]["kernel"], round(d["roofline"]["frac"],4))
PY
done
python -c "
import json; d=json.loads(open('$O/bench_c5.json').read()); print('c5', d['ms_per_step'], d['cg_iterations_per_step'])"
head -12 gpurun_out/prof_r02e/summary.txt
