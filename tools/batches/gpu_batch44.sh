#!/bin/bash
timeout 1500 python tools/slab_soak.py 3000 3 2>&1 | tail -4
