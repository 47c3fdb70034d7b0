# time of the two WCSPH neighbour passes against the number of workgroups a CU can hold (unused dynamic LDS takes the room away)
O=gpurun_out/r03o; mkdir -p $O
A="--steps 30 --warmup 5 --no-cpu-baseline --no-extras --motion-step 0 --all-kernels"
for x in 0 9000 16000 30000 42000 56000 90000; do
  SPH_DEBUG_EXTRA_LDS=$x timeout -s KILL 90 python bench.py $A > $O/x$x.json 2> $O/x$x.err
  echo "extra LDS $x: $(grep -h 'density \|wcsph_forces' $O/x$x.err | tr -s ' ' | tr '\n' ';')"
done
