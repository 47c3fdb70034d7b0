#!/bin/bash
O=gpurun_out/b31; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_round2.py -m gpu -q -k "advance or driver or end_to_end" > $O/tests.log 2>&1; tail -5 $O/tests.log | cut -c1-300
python tools/step_overhead.py 2>&1 | tail -3
