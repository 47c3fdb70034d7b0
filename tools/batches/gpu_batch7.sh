#!/bin/bash
O=gpurun_out/b7; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q --durations=6 > $O/tests.log 2>&1; tail -30 $O/tests.log
