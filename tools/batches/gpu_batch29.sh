#!/bin/bash
O=gpurun_out/b29; mkdir -p $O
SPH_TEST_HUGE=1 timeout 1500 python -m pytest tests/test_hip_round2.py -m gpu -q -s -k "million or capacity" --durations=3 > $O/tests.log 2>&1; grep "particles:\|passed\|failed\|Error\|error\|assert\|s call" $O/tests.log | cut -c1-300 | head
