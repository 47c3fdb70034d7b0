#!/bin/bash
O=gpurun_out/b23; mkdir -p $O
timeout 900 python tools/long_run.py > $O/long_run.log 2>&1; tail -15 $O/long_run.log
