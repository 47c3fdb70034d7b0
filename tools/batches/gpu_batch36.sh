#!/bin/bash
O=gpurun_out/b36; mkdir -p $O
for k in 1 2 3; do timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests$k.log 2>&1; tail -1 $O/tests$k.log; done
