# tile experiment: bitwise comparison against k_nbr_pass, parity tests under SPH_TILE=1, timing.  Every command under a short timeout.
O=gpurun_out/r03t; mkdir -p $O
T="timeout -s KILL 60"
$T python tools/debug_state.py c1 40 0 old0 2>/dev/null | tail -1
SPH_TILE=1 $T python tools/debug_state.py c1 40 0 tile0 2>/dev/null | tail -1; echo "rc=$?"
python - <<'PY'
import numpy as np, os
if os.path.exists("gpurun_out/state_c1_tile0.npz"):
    a=np.load("gpurun_out/state_c1_old0.npz"); b=np.load("gpurun_out/state_c1_tile0.npz")
    print("c1 strict", {k: bool(np.array_equal(a[k], b[k])) for k in a.files}, "max |dx|", float(np.abs(a["x"]-b["x"]).max()))
PY
[ -f gpurun_out/state_c1_tile0.npz ] || exit 1
A="--steps 100 --warmup 10 --no-cpu-baseline --no-extras --all-kernels"
timeout -s KILL 120 python bench.py $A > $O/old.json 2> $O/old.err; SPH_TILE=1 timeout -s KILL 120 python bench.py $A > $O/tile.json 2> $O/tile.err
grep -h "density \|wcsph_forces\|block_prep\|misc" $O/old.err $O/tile.err
python - <<'PY'
import json
for v in ("old","tile"):
    try:
        d=json.loads(open(f"gpurun_out/r03t/{v}.json").read().strip().split("\n")[-1]); print(v, "%.4f rest %.4f motion" % (d["ms_per_step"], d["in_motion"]["ms_per_step"]), "fallback rest", d["config"]["lds_fallback_blocks_last_step"], "motion", d["in_motion"]["lds_fallback_blocks_last_step"])
    except Exception as e: print(v, "failed", e)
PY
SPH_TILE=1 timeout -s KILL 240 python -m pytest tests/test_hip_wcsph.py tests/test_big_golden.py -m gpu -x -q 2>&1 | tail -3
