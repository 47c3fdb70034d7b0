#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/tr_c2
timeout 200 rocprofv3 --kernel-trace -d /tmp/tr_c2 -o trace --output-format csv -- python $R/bench.py --steps 20 --warmup 3 --repeats 1 --no-cpu-baseline --no-extras --motion-step 0 > /dev/null 2>&1
python $R/tools/trace_gaps.py /tmp/tr_c2 --tail-ms 1e9 --max-gap-us 40
