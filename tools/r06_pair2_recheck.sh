cd ${GRAFT_REPO_ROOT:-.}
V=sph_project_amd/variants/libsph_hip_nopair2.so
for v in "X=1" "SPH_HIP_LIB=$V" "X=1" "SPH_HIP_LIB=$V"; do
  env $v python bench.py --config c3 --steps 50 --warmup 5 --no-cpu-baseline --no-extras --motion-step 1000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C3 2+2 [%-60s] %.4f ms/step from rest, %.4f at step 1000' % ('$v', d['ms_per_step'], d['in_motion']['ms_per_step']))"
  env $v python bench.py --config c3 --measured-iterations --presteps 1000 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --motion-step 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C3 measured, step 1000+ [%-44s] %.4f ms/step' % ('$v', d['ms_per_step']))"
  env $v python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); c=d['extras']['c3']; print('driver extras [%-40s] c3 %.4f  at 1000: %.4f  measured rest %.4f  motion %.4f' % ('$v', c['ms_per_step'], c['in_motion']['ms_per_step'], c['measured']['from_rest']['ms_per_step'], c['measured']['in_motion']['ms_per_step']))"
done
