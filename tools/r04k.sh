O=gpurun_out/r04k; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/gpu_suite.txt 2>&1; tail -4 $O/gpu_suite.txt
timeout 200 python -m pytest tests/test_hip_rigid.py -q -s -k wrench 2>&1 | grep -h "^wrench:\|passed\|failed"
BENCH_ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-extras --all-kernels" timeout 400 tools/ab.sh r04k base="" l7="SPH_HIP_LIB=sph_project_amd/variants/libsph_hip_l7.so" base2="" l7b="SPH_HIP_LIB=sph_project_amd/variants/libsph_hip_l7.so"
