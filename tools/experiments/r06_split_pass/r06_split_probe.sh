#!/bin/bash
# Round 6: what does ONE rank's share of a strong-scaled scene cost (tools/slab_size_probe.py: unsharded blocks shaped like a slab of C4 / 8,
# C2 / 8 and C2 / 2) with the density and force passes split three ways per tile (SplitPass; SPH_SPLIT_TILES = tiles below which a pass is
# split) and with the scan's tile sums left by the hashers (SPH_NO_SCAN_FOLD=1: two scan launches as before)?
cd ${GRAFT_REPO_ROOT:-.}
for v in "SPH_SPLIT_TILES=0 SPH_NO_SCAN_FOLD=1" "SPH_SPLIT_TILES=0" "SPH_SPLIT_TILES=1536" "SPH_SPLIT_TILES=4096" "SPH_SPLIT_TILES=0 SPH_NO_SCAN_FOLD=1" "SPH_SPLIT_TILES=4096"; do
  echo "== $v"
  env $v python tools/slab_size_probe.py --steps 300 2>/dev/null
done
