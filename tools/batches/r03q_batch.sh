# light passes at 6 workgroups per CU with the ordered path compiled out (timing experiment: groups that need that path are computed wrongly)
O=gpurun_out/r03q; mkdir -p $O
A="--steps 100 --warmup 10 --no-cpu-baseline --no-extras --all-kernels"
for v in main lean6; do
  if [ $v = main ]; then unset SPH_HIP_LIB; else export SPH_HIP_LIB=$PWD/sph_project_amd/variants/libsph_hip_$v.so; fi
  timeout -s KILL 150 python bench.py $A > $O/c2_$v.json 2> $O/c2_$v.err
  timeout -s KILL 150 python bench.py $A --config c3 --motion-step 0 > $O/c3_$v.json 2> $O/c3_$v.err
  echo "$v: $(grep -h 'density \|wcsph_forces\|dfsph_density_alpha\|dfsph_rho_adv\|dfsph_correct' $O/c2_$v.err $O/c3_$v.err | tr -s ' ' | tr '\n' ';')"
done
unset SPH_HIP_LIB
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03q/*.json")):
    try:
        d=json.loads(open(f).read().strip().split("\n")[-1]); im=d.get("in_motion") or {}
        print(f.split("/")[-1].ljust(16), "%.4f rest" % d["ms_per_step"], ("%.4f motion" % im["ms_per_step"]) if im else "")
    except Exception as e: print(f, "failed", e)
PY
