#!/bin/bash
# same-box A/B of the tile order of the neighbour passes (Consts::xcd_chunk, SPH_XCD_CHUNK): 0 = one contiguous eighth per XCD
cd ${GRAFT_REPO_ROOT:-.}
tools/ab.sh r06_xcd_chunk base="" c8="SPH_XCD_CHUNK=8" c16="SPH_XCD_CHUNK=16" c32="SPH_XCD_CHUNK=32" c64="SPH_XCD_CHUNK=64" c128="SPH_XCD_CHUNK=128" c1="SPH_XCD_CHUNK=1" base2="" c32b="SPH_XCD_CHUNK=32" 2>&1 | grep -v "^    " | tee gpurun_out/r06_xcd_chunk/summary.txt
