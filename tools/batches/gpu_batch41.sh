#!/bin/bash
python bench.py --no-cpu-baseline --config c3 --motion-step 0 --all-kernels 2>&1 >/dev/null | grep -v "No rigid" | head -12
echo ---- in motion
python bench.py --no-cpu-baseline --config c3 --motion-step 0 --presteps 2500 --all-kernels 2>&1 >/dev/null | grep -v "No rigid" | head -12
