#!/bin/bash
# GPU batch 5 (round 2): AoS phase 1 with 4-candidate chunks + 5 workgroups per CU for the light functors, exact acceptance
# threshold, merged reciprocals: full suite + bench + c3/c4 lines
O=gpurun_out/b5; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q --durations=6 > $O/tests.log 2>&1; tail -25 $O/tests.log
timeout 600 python bench.py --all-kernels > $O/bench_c2.json 2> $O/bench_c2.err
timeout 600 python bench.py --config c3 --no-cpu-baseline --all-kernels > $O/bench_c3.json 2> $O/bench_c3.err
timeout 600 python bench.py --config c4 --no-cpu-baseline --all-kernels > $O/bench_c4.json 2> $O/bench_c4.err
timeout 900 python tools/bench_c5.py > $O/bench_c5.json 2> $O/bench_c5.err
python - <<'PY'
import json
for c in ("c2","c3","c4"):
    d=json.loads(open(f"gpurun_out/b5/bench_{c}.json").read().strip().splitlines()[-1])
    print(c, "%.4f ms" % d["ms_per_step"], ["%.4f" % x for x in d["repeat_ms_per_step"]], "motion", d["in_motion"] and "%.4f" % d["in_motion"]["ms_per_step"], d["roofline"]["kernel"], "%.1f us" % d["roofline"]["avg_launch_us"])
    print(open(f"gpurun_out/b5/bench_{c}.err").read()[:420])
print(open("gpurun_out/b5/bench_c5.json").read()[:600])
PY
