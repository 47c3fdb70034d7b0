#!/bin/bash
python tools/step_overhead.py 2>&1 | tail -1
