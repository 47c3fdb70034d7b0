# a fifth workgroup per CU for the wide-record functors at the price of spills outside the pair loops: A/B
O=gpurun_out/r03p5; mkdir -p $O
A="--steps 100 --warmup 10 --no-cpu-baseline --no-extras --all-kernels"
for v in main h5sb1 h5sb2; do
  if [ $v = main ]; then unset SPH_HIP_LIB; else export SPH_HIP_LIB=$PWD/sph_project_amd/variants/libsph_hip_$v.so; fi
  timeout -s KILL 150 python bench.py $A > $O/c2_$v.json 2> $O/c2_$v.err
  timeout -s KILL 150 python bench.py $A --config c3 > $O/c3_$v.json 2> $O/c3_$v.err
  echo "$v: $(grep -h 'wcsph_forces\|non_pressure\|dfsph_density_alpha' $O/c2_$v.err $O/c3_$v.err | tr -s ' ' | tr '\n' ';')"
done
unset SPH_HIP_LIB
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03p5/*.json")):
    try:
        d=json.loads(open(f).read().strip().split("\n")[-1]); print(f.split("/")[-1].ljust(16), "%.4f rest %.4f motion" % (d["ms_per_step"], d["in_motion"]["ms_per_step"]), "fallback", d["in_motion"]["lds_fallback_blocks_last_step"])
    except Exception as e: print(f, "failed", e)
PY
