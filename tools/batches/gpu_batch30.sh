#!/bin/bash
O=gpurun_out/b30; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_wcsph.py -m gpu -q -s -k "piled" > $O/tests.log 2>&1; grep "per cell\|passed\|failed\|Error\|error\|assert\|Mismatch" $O/tests.log | cut -c1-400 | head -20
