T=r03i; mkdir -p gpurun_out/$T; O=$PWD/gpurun_out/$T; R=$PWD
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
python tools/scene0_iterations.py --steps 60 --scale 0.25 --scale-steps 12 > $O/scene0_iterations.json 2> $O/scene0.err
python - $O <<'PY'
import json,sys
d=json.load(open(sys.argv[1]+'/scene0_iterations.json'))
f=d['full_size']; print('scene0 without meshes, full size:', f['particles'], 'ms/step', round(f['ms_per_step'],3), 'iters/step', f['solver_iterations_per_step'])
s=d['scaled_copy']; print('scaled copy:', s['particles'], 'hip', s['hip_iterations_div_den'], 'oracle', s['oracle_iterations_div_den'], 'maxdiff', s['max_difference'])
PY
( cd _refdata && for s in final_scene0 final_scene1 final_scene2 final_scene3 final_scene4 dragon_bath_dfsph dragon_bath_wcsph dragon_bath_pcisph high_fluid_dfsph high_fluid_pcisph test; do
  timeout 600 python $R/sph_project_amd/run_simulation.py --scene_file data/scenes/$s.json --max_steps 60 --output_dir $O/out_$s > $O/$s.log 2>&1
  echo "$s rc=$? $(grep -E 'Simulation Finished|Error|error' $O/$s.log | tail -1 | cut -c1-220)"
  rm -rf $O/out_$s
done ) > $O/reference_scenes.txt 2>&1
cat $O/reference_scenes.txt
bash tools/prof.sh r03 > $O/prof.log 2>&1; tail -5 $O/prof.log
python tools/prof_summary.py gpurun_out/prof_r03 --json gpurun_out/prof_r03/pmc_derived.json --config c2 --source profiles/r03_rocprofv3_c2_summary.txt > gpurun_out/prof_r03/summary.txt 2>&1
rm -rf gpurun_out/prof_r03/trace gpurun_out/prof_r03/pmc1 gpurun_out/prof_r03/pmc2 gpurun_out/prof_r03/pmc3 gpurun_out/prof_r03/pmc4
head -24 gpurun_out/prof_r03/summary.txt
