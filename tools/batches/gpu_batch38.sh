#!/bin/bash
O=gpurun_out/b38; mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_slab.py tests/test_hip_round2.py -m gpu -q -k "rigid or rccl or spawns or rebalanc or cuts" > $O/tests.log 2>&1; tail -3 $O/tests.log | cut -c1-200
