#!/bin/bash
O=gpurun_out/b29; mkdir -p $O
free -g | head -2
timeout 900 python -m pytest tests/test_hip_round2.py -m gpu -q -s -k "64_million" > $O/tests.log 2>&1; grep "64 M\|passed\|failed\|Error\|error\|assert" $O/tests.log | cut -c1-300 | head
