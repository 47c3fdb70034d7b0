O=gpurun_out/r04h; mkdir -p $O
timeout 300 python -m pytest tests/test_hip_rigid.py -x -q -s > $O/rigid.txt 2>&1; grep -h "wrench:\|passed\|failed\|Error" $O/rigid.txt | tail -5
BENCH_ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-extras --all-kernels" timeout 300 tools/ab.sh r04h perm="" noperm="SPH_NO_LANE_PERM=1" perm2="" noperm2="SPH_NO_LANE_PERM=1"
