T=r03f; mkdir -p gpurun_out/$T; O=gpurun_out/$T
A="--steps 100 --warmup 10 --no-cpu-baseline --no-extras"
for rep in 1 2; do
python bench.py $A > $O/c2_main_$rep.json 2>/dev/null
SPH_HIP_LIB=$(pwd)/sph_project_amd/variants/libsph_hip_p2x2.so python bench.py $A > $O/c2_p2x2_$rep.json 2>/dev/null
done
python bench.py $A --config c3 > $O/c3_main.json 2>/dev/null
SPH_HIP_LIB=$(pwd)/sph_project_amd/variants/libsph_hip_p2x2.so python bench.py $A --config c3 > $O/c3_p2x2.json 2>/dev/null
python - $O <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/c*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1].ljust(20), '%.4f ms/step rest' % d['ms_per_step'], ('%.4f in motion' % d['in_motion']['ms_per_step']) if d.get('in_motion') else '', d['roofline']['kernel'], '%.1f us' % d['roofline']['avg_launch_us'])
    except Exception as e: print(f,'ERR',e)
PY
python tools/scene0_iterations.py --steps 60 --scale 0.25 --scale-steps 12 > $O/scene0_iterations.json 2> $O/scene0.err; tail -c 400 $O/scene0.err
python - $O <<'PY'
import json,sys
d=json.load(open(sys.argv[1]+'/scene0_iterations.json'))
f=d['full_size']; print('scene0 full:', f['particles'], 'ms/step', round(f['ms_per_step'],2), 'iters/step', f['solver_iterations_per_step'], 'ms/iter', round(f['ms_per_solver_iteration'],4))
print([ (r['iter_divergence'], r['iter_density']) for r in f['per_step'][:20]])
s=d['scaled_copy']; print('scaled:', s['particles'], s['hip_iterations_div_den'], s['oracle_iterations_div_den'], 'maxdiff', s['max_difference'], 'oracle s', s['oracle_seconds'])
PY
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
