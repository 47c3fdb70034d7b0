#!/bin/bash
O=gpurun_out/b43; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_round2.py -m gpu -q -s -k "release_their_memory" > $O/tests.log 2>&1; grep "lost\|passed\|failed\|Error\|assert" $O/tests.log | cut -c1-300 | head
