# same box A/B: payload-free passes at 6 (committed) vs 5 workgroups per CU, both without the L2 walk
O=gpurun_out/r03u; mkdir -p $O
A="--steps 100 --warmup 10 --no-cpu-baseline --no-extras --all-kernels"
for v in main light5 main light5; do
  if [ $v = main ]; then unset SPH_HIP_LIB; else export SPH_HIP_LIB=$PWD/sph_project_amd/variants/libsph_hip_$v.so; fi
  timeout -s KILL 150 python bench.py $A --motion-step 0 > $O/c2_$v.json 2> $O/c2_$v.err
  echo "$v: $(grep -h 'density \|wcsph_forces' $O/c2_$v.err | tr -s ' ' | sed 's/launches [0-9]* avg//' | tr '\n' ';') $(python -c "import json;print(json.loads(open('$O/c2_$v.json').read().strip().split(chr(10))[-1])['ms_per_step'])")"
done
