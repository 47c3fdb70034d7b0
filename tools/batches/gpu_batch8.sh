#!/bin/bash
O=gpurun_out/b8; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; tail -8 $O/tests.log
bash tools/prof.sh r02b --repeats 1 --motion-step 0 > $O/prof_rest.log 2>&1; tail -5 $O/prof_rest.log
SKIP_TRACE= PMC_SETS="1 2" bash tools/prof.sh r02b_motion --repeats 1 --motion-step 0 --presteps 2500 > $O/prof_motion.log 2>&1; tail -3 $O/prof_motion.log
