#!/bin/bash
# manual check (needs a local, untracked copy of the reference's data/ directory under _refdata/): the reference's own
# scene files, unmodified, through the drop-in driver
O=$PWD/gpurun_out/b34; mkdir -p $O
cd _refdata
for s in final_scene3 final_scene0 final_scene2 dragon_bath_dfsph dragon_bath_wcsph dragon_bath_pcisph final_scene1 final_scene4 high_fluid_wcsph high_fluid_dfsph high_fluid_pcisph test; do
  timeout 600 python ../sph_project_amd/run_simulation.py --scene_file data/scenes/$s.json --max_steps 60 --output_dir $O/out_$s > $O/$s.log 2>&1
  echo "$s rc=$? $(grep -E 'Simulation Finished|Error|error' $O/$s.log | tail -2 | cut -c1-200)"
  rm -rf $O/out_$s
done
