#!/bin/bash
O=gpurun_out/b27; mkdir -p $O
SPH_HIP_LIB=$PWD/sph_project_amd/variants/libsph_hip_tl.so SPH_DEBUG_MODE=20 python bench.py --no-cpu-baseline --motion-step 0 --steps 12 --warmup 3 --repeats 1 > $O/tl.json 2> $O/tl.err
grep timeline $O/tl.err | head -12
