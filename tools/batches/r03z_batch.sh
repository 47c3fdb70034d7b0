# decomposition by removal on the committed build + lane permutation A/B (with the prologue now one round trip behind the permutation load)
O=gpurun_out/r03z; mkdir -p $O
A="--steps 10 --warmup 2 --no-cpu-baseline --no-extras --motion-step 0 --all-kernels"
for m in 11 12; do SPH_DEBUG_MODE=$m python bench.py $A > $O/mode$m.json 2> $O/mode$m.err; echo "mode $m"; grep -h "density\|wcsph_forces" $O/mode$m.err; done
B="--steps 100 --warmup 10 --no-cpu-baseline --no-extras --all-kernels"
python bench.py $B > $O/perm.json 2> $O/perm.err; echo perm; grep -h "density\|wcsph_forces" $O/perm.err
SPH_NO_LANE_PERM=1 python bench.py $B > $O/noperm.json 2> $O/noperm.err; echo noperm; grep -h "density\|wcsph_forces" $O/noperm.err
python - <<'PY'
import json
for v in ("perm","noperm"):
    d=json.loads(open(f"gpurun_out/r03z/{v}.json").read().strip().split("\n")[-1]); print(v, "%.4f rest %.4f motion" % (d["ms_per_step"], d["in_motion"]["ms_per_step"]))
PY
