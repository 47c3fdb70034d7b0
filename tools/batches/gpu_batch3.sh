#!/bin/bash
# GPU batch 3 (round 2): full GPU suite on the new build (no-SLP, coalesced scan, per-functor instantiations), bench, w5 A/B
O=gpurun_out/b3; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q --durations=12 > $O/tests.log 2>&1; tail -45 $O/tests.log
timeout 600 python bench.py --all-kernels --no-cpu-baseline > $O/bench_base.json 2> $O/bench_base.err
SPH_HIP_LIB=$PWD/sph_project_amd/variants/libsph_hip_w5.so timeout 600 python bench.py --all-kernels --no-cpu-baseline > $O/bench_w5.json 2> $O/bench_w5.err
for f in $O/bench_base $O/bench_w5; do python - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]+".json").read().strip().splitlines()[-1])
print(sys.argv[1], "rest %.4f ms" % d["ms_per_step"], d["repeat_ms_per_step"], "motion %.4f" % d["in_motion"]["ms_per_step"], d["roofline"]["kernel"], "%.1f us" % d["roofline"]["avg_launch_us"])
print(open(sys.argv[1]+".err").read()[:600])
PY
done
