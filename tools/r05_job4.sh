#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD
timeout 120 tools/mb/mb_gridbar > gpurun_out/r05_mb_gridbar.txt 2>&1
python -m pytest tests/test_hip_wcsph.py tests/test_hip_golden.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r05_scatter_tests.txt
python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras --all-kernels > gpurun_out/r05_scatter_bench.json 2> gpurun_out/r05_scatter_bench.err
export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/tr_c3 /tmp/tr_c5 /tmp/tr_c2
timeout 200 rocprofv3 --kernel-trace -d /tmp/tr_c3 -o trace --output-format csv -- python $R/bench.py --config c3 --measured-iterations --presteps 1000 --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-extras --motion-step 0 > /dev/null 2>&1
python $R/tools/trace_gaps.py /tmp/tr_c3 --tail-ms 60 > $R/gpurun_out/r05_gaps_c3_motion.txt 2>&1
timeout 200 rocprofv3 --kernel-trace -d /tmp/tr_c5 -o trace --output-format csv -- python $R/tools/bench_c5.py --no-events --steps 10 --warmup 2 > /dev/null 2>&1
python $R/tools/trace_gaps.py /tmp/tr_c5 --tail-ms 15 > $R/gpurun_out/r05_gaps_c5.txt 2>&1
timeout 200 rocprofv3 --kernel-trace -d /tmp/tr_c2 -o trace --output-format csv -- python $R/bench.py --steps 20 --warmup 3 --repeats 1 --no-cpu-baseline --no-extras --motion-step 0 > /dev/null 2>&1
python $R/tools/trace_gaps.py /tmp/tr_c2 --tail-ms 5 > $R/gpurun_out/r05_gaps_c2.txt 2>&1
cd $R; cat gpurun_out/r05_mb_gridbar.txt gpurun_out/r05_scatter_tests.txt gpurun_out/r05_gaps_c3_motion.txt gpurun_out/r05_gaps_c5.txt gpurun_out/r05_gaps_c2.txt
