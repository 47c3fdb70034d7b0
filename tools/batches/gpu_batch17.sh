#!/bin/bash
O=gpurun_out/b17; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_golden.py -m gpu -q -s > $O/tests.log 2>&1; grep "npz\|passed\|failed\|Error" $O/tests.log | cut -c1-200 | head -70
