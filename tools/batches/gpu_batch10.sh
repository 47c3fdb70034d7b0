#!/bin/bash
O=gpurun_out/b10; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --durations=5 > $O/tests.log 2>&1; tail -40 $O/tests.log
