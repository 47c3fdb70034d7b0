# round 3, final: smoke, full GPU suite, kernel trace + PMC passes, the driver's default bench line
O=gpurun_out/r03final; mkdir -p $O
timeout -s KILL 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -s KILL 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout -s KILL 600 bash tools/prof.sh r03final --motion-step 0 > $O/prof.log 2>&1; echo "prof rc=$?"
timeout -s KILL 600 python bench.py > $O/bench_driver_line.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03final/bench_driver_line.json").read().strip().split("\n")[-1])
r=d["roofline"]; print({k:d[k] for k in ("value","ms_per_step")}, r["kernel"], round(r["avg_launch_us"],1), round(r["frac"],4), "runner_up", r.get("runner_up",{}).get("kernel"), round(r.get("runner_up",{}).get("frac",0),4), "step_frac", round(r["step_frac"],4))
print("in_motion", d["in_motion"]["ms_per_step"], "extras", {k:v.get("ms_per_step") for k,v in d.get("extras",{}).items()}, "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
