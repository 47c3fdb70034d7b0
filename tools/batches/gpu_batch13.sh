#!/bin/bash
O=gpurun_out/b13; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --durations=8 > $O/tests.log 2>&1; tail -25 $O/tests.log | cut -c1-300
