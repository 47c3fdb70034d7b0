cd ${GRAFT_REPO_ROOT:-.}
tools/ab.sh r06_knobs base="" hsb4="SPH_HIP_LIB=sph_project_amd/variants/libsph_hip_hsb4.so" base2="" hsb4b="SPH_HIP_LIB=sph_project_amd/variants/libsph_hip_hsb4.so" 2>&1 | grep -v "^    "
for v in "X=1" "SPH_HIP_LIB=sph_project_amd/variants/libsph_hip_msb3.so" "X=1" "SPH_HIP_LIB=sph_project_amd/variants/libsph_hip_msb3.so"; do
  env $v python bench.py --config c3 --steps 50 --warmup 5 --no-cpu-baseline --no-extras --motion-step 1000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C3 2+2 [%-60s] %.4f ms/step from rest, %.4f at step 1000' % ('$v', d['ms_per_step'], d['in_motion']['ms_per_step']))"
done
python -m pytest tests/test_hip_wcsph.py tests/test_hip_golden.py -m gpu -x -q 2>&1 | tail -2
