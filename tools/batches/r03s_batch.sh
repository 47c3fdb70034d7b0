# L2 walk replaced by chunked staging of oversized runs; light passes at 6 workgroups per CU: correctness, then timing
O=gpurun_out/r03s; mkdir -p $O
timeout -s KILL 500 python -m pytest tests/test_hip_wcsph.py tests/test_hip_golden.py tests/test_big_golden.py tests/test_hip_solvers.py tests/test_hip_rigid.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
A="--steps 100 --warmup 10 --no-cpu-baseline --no-extras --all-kernels"
timeout -s KILL 150 python bench.py $A > $O/c2.json 2> $O/c2.err
timeout -s KILL 150 python bench.py $A --config c3 > $O/c3.json 2> $O/c3.err
timeout -s KILL 150 python bench.py $A --method pcisph --motion-step 0 > $O/pci.json 2> $O/pci.err
echo "$(grep -h 'density \|wcsph_forces\|dfsph_density_alpha\|dfsph_rho_adv\|dfsph_correct\|non_pressure' $O/c2.err $O/c3.err | tr -s ' ' | sed 's/launches [0-9]* avg//' | tr '\n' ';')"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03s/*.json")):
    try:
        d=json.loads(open(f).read().strip().split("\n")[-1]); im=d.get("in_motion") or {}
        print(f.split("/")[-1].ljust(12), "%.4f rest" % d["ms_per_step"], ("%.4f motion, fallback %d" % (im["ms_per_step"], im["lds_fallback_blocks_last_step"])) if im else "")
    except Exception as e: print(f, "failed", e)
PY
