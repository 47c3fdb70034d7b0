# lean / full split of the neighbour passes: correctness subset, then A/B against SPH_NO_LEAN=1 (same library)
O=gpurun_out/r03r; mkdir -p $O
timeout -s KILL 400 python -m pytest tests/test_hip_wcsph.py tests/test_hip_golden.py tests/test_big_golden.py tests/test_hip_solvers.py tests/test_hip_rigid.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
A="--steps 100 --warmup 10 --no-cpu-baseline --no-extras --all-kernels"
for v in lean nolean; do
  if [ $v = lean ]; then unset SPH_NO_LEAN; else export SPH_NO_LEAN=1; fi
  timeout -s KILL 150 python bench.py $A > $O/c2_$v.json 2> $O/c2_$v.err
  timeout -s KILL 150 python bench.py $A --config c3 > $O/c3_$v.json 2> $O/c3_$v.err
  timeout -s KILL 150 python bench.py $A --method pcisph --motion-step 0 > $O/pci_$v.json 2> $O/pci_$v.err
  echo "$v: $(grep -h 'density \|wcsph_forces\|dfsph_density_alpha\|dfsph_rho_adv\|dfsph_correct\|non_pressure' $O/c2_$v.err $O/c3_$v.err | tr -s ' ' | sed 's/launches [0-9]* avg//' | tr '\n' ';')"
done
unset SPH_NO_LEAN
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03r/*.json")):
    try:
        d=json.loads(open(f).read().strip().split("\n")[-1]); im=d.get("in_motion") or {}
        print(f.split("/")[-1].ljust(16), "%.4f rest" % d["ms_per_step"], ("%.4f motion, fallback %d" % (im["ms_per_step"], im["lds_fallback_blocks_last_step"])) if im else "")
    except Exception as e: print(f, "failed", e)
PY
