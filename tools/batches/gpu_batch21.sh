#!/bin/bash
O=gpurun_out/b21; mkdir -p $O
python bench.py --no-cpu-baseline --all-kernels --motion-step 0 > $O/rest.json 2> $O/rest.err
python bench.py --no-cpu-baseline --all-kernels --presteps 2500 --motion-step 0 > $O/motion.json 2> $O/motion.err
grep -v "No rigid" $O/rest.err | head -30; echo ----; grep -v "No rigid" $O/motion.err | head -30
