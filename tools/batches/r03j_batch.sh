O=$PWD/gpurun_out/r03j; mkdir -p $O
python tools/scene0_iterations.py --skip-full --scale 0.25 --scale-steps 12 > $O/scaled.json 2> $O/scaled.err; tail -c 300 $O/scaled.err
python - $O <<'PY'
import json,sys
s=json.load(open(sys.argv[1]+'/scaled.json'))['scaled_copy']
print('hip   ', s['hip_iterations_div_den']); print('oracle', s['oracle_iterations_div_den']); print('drift ', ['%.1e'%d for d in s['drift_vs_oracle_per_step']])
print('errs (div hip, div oracle, den hip, den oracle)', [tuple('%.3e'%v for v in e) for e in s['final_errors_hip_oracle_div_den']])
PY
bash tools/prof.sh r03 --motion-step 0 > $O/prof.log 2>&1
python tools/prof_summary.py gpurun_out/prof_r03 --json gpurun_out/prof_r03/pmc_derived.json --config c2 --source profiles/r03_rocprofv3_c2_summary.txt > gpurun_out/prof_r03/summary.txt 2>&1
rm -rf gpurun_out/prof_r03/trace gpurun_out/prof_r03/pmc1 gpurun_out/prof_r03/pmc2 gpurun_out/prof_r03/pmc3 gpurun_out/prof_r03/pmc4
head -16 gpurun_out/prof_r03/summary.txt; tail -8 gpurun_out/prof_r03/summary.txt
