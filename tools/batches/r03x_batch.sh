O=gpurun_out/r03x; mkdir -p $O
for v in base main nopre; do
  if [ $v = main ]; then unset SPH_HIP_LIB; else export SPH_HIP_LIB=$PWD/sph_project_amd/variants/libsph_hip_$v.so; fi
  for f in 0 1; do python tools/debug_alpha.py dfsph_implicit_box $f 2>/dev/null | sed "s/^/$v fast=$f /"; done
done > $O/alpha.txt
unset SPH_HIP_LIB
python - <<'PY' >> gpurun_out/r03x/alpha.txt
import numpy as np
for f in (0, 1):
    a = np.load(f"gpurun_out/debug_alpha_dfsph_implicit_box_base_{f}.npz"); 
    for v in ("main", "nopre"):
        b = np.load(f"gpurun_out/debug_alpha_dfsph_implicit_box_{v}_{f}.npz")
        print("fast" if f else "strict", v, "vs base: bitwise equal fields:", {k: bool(np.array_equal(a[k], b[k])) for k in a.files})
PY
cat $O/alpha.txt
