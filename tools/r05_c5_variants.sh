#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for v in new old new old new old; do
  if [ $v = old ]; then export SPH_HIP_LIB=$PWD/sph_project_amd/variants/libsph_hip_noearly.so; else unset SPH_HIP_LIB; fi
  python tools/bench_c5.py --no-events --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C5 $v (no events): %.4f ms/step %.1f CG it' % (d['ms_per_step'], d['cg_iterations_per_step']))"
done
for v in new old new old; do
  if [ $v = old ]; then export SPH_HIP_LIB=$PWD/sph_project_amd/variants/libsph_hip_noearly.so; else unset SPH_HIP_LIB; fi
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras --motion-step 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C2 $v %.4f ms/step' % (d['ms_per_step']))"
  python bench.py --config c3 --steps 50 --warmup 5 --no-cpu-baseline --no-extras --motion-step 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C3 2+2 $v %.4f ms/step' % d['ms_per_step'])"
done
