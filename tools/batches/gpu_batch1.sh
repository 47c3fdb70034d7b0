#!/bin/bash
# GPU batch 1 (round 2): issue-rate microbenchmarks, baseline tests, A/B of -fno-slp-vectorize
O=gpurun_out/b1; mkdir -p $O
( cd tools/mb && timeout 300 ./mb_issue > ../../$O/mb_issue.txt 2>&1; timeout 300 ./mb_issue_noslp > ../../$O/mb_issue_noslp.txt 2>&1 )
timeout 600 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
for v in base noslp; do
  if [ $v = base ]; then unset SPH_HIP_LIB; else export SPH_HIP_LIB=$PWD/sph_project_amd/variants/libsph_hip_$v.so; fi
  timeout 300 python bench.py --no-cpu-baseline --steps 200 --warmup 20 --all-kernels > $O/bench_${v}_rest.json 2> $O/bench_${v}_rest.err
  timeout 300 python bench.py --no-cpu-baseline --steps 200 --warmup 20 --presteps 2500 --all-kernels > $O/bench_${v}_motion.json 2> $O/bench_${v}_motion.err
done
grep -h ms_per_step $O/bench_*.json | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_us'])
"
