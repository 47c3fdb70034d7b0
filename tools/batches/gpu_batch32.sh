#!/bin/bash
O=gpurun_out/b32; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_slab.py -m gpu -q -s -k "implicit" > $O/tests.log 2>&1; grep "implicit \|passed\|failed\|Error\|assert" $O/tests.log | cut -c1-400 | head
