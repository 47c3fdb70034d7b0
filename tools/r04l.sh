O=gpurun_out/r04l; mkdir -p $O
timeout 700 python -m pytest tests -m gpu -q > $O/gpu_suite.txt 2>&1; tail -6 $O/gpu_suite.txt
grep -h "^wrench:" $O/gpu_suite.txt
timeout 200 python -m pytest tests/test_hip_rigid.py -q -s -k wrench 2>&1 | grep -h "^wrench:\|passed\|failed"
