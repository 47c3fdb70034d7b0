#!/bin/bash
O=gpurun_out/b26; mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_slab.py tests/test_hip_rigid.py -m gpu -q -s -k "rigid or slab_sharding_matches" > $O/tests.log 2>&1; grep "dynamic rigid\|passed\|failed\|Error\|error\|assert" $O/tests.log | cut -c1-400 | head -30
