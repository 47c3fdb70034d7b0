#!/bin/bash
# per-kernel timing with debug modes of the neighbour kernel (0 normal, 2 phase-1 only, 3 staging only)
for mode in ${MODES:-0 2 3}; do
  echo "== force_global/debug mode $mode"
  python - <<PY 2>&1 | grep -v "No rigid"
import sys; sys.path.insert(0,'.')
from sph_project_amd import product as bench
from tests import helpers as H
cfg = bench.c2_scene()
c, s = H.build_product(cfg, fast_math=1, force_global=$mode)
s.prepare(); e = c.engine
e.step_async(3); e.synchronize()
e.profile_enable(-1, True); e.profile_reset(); e.step_async(5); e.synchronize()
for k in range(18):
    n, ms = e.profile_read(k)
    if n: print("  %-22s %8.1f us" % (e.lib.sph_kernel_name(k).decode(), 1e3*ms/n))
st = s.stats(); print("  lds_fallback runs last step:", st["lds_fallback_blocks"], " pairs:", st["pair_interactions"])
PY
done
