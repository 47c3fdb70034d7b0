O=$PWD/gpurun_out/r03g; mkdir -p $O; R=$PWD
cd _refdata
for s in final_scene0 final_scene4; do
python $R/tools/scene0_iterations.py --scene-file data/scenes/$s.json --steps 30 > $O/$s.json 2> $O/$s.err
python - $O/$s.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(d['scene'], d['particles'], d['fluid_particles'], 'ms/step %.2f' % d['ms_per_step'], 'iters/step', d['solver_iterations_per_step'])
print(' per step (div, den, ms, fallback):', [(r['iter_divergence'], r['iter_density'], round(r['ms'],1), r['lds_fallback_blocks']) for r in d['per_step'][:12]])
print(' kernels [launches/step, ms/step]:', d['kernels_ms_per_step'])
PY
done
