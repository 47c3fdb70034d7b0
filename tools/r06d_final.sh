# the closing run of round 6, fourth session (after the sort by run lists): same recipe as tools/r06_final.sh
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r06d_final; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/gpu_suite.txt 2>&1; tail -4 $O/gpu_suite.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_line.json 2> $O/bench_driver_line.err; tail -c 300 $O/bench_driver_line.json; echo
cp profiles/pmc_derived.json gpurun_out/pmc_derived.json
for t in c2 c3 c5; do
  rm -rf gpurun_out/prof_r06g_$t
  case $t in
    c2) JSON_ARGS="--json gpurun_out/pmc_derived.json --config c2 --source profiles/r06d_final_rocprofv3_c2_summary.txt" bash tools/prof.sh r06g_c2 > $O/prof_c2.log 2>&1;;
    c3) JSON_ARGS="--json gpurun_out/pmc_derived.json --config c3 --source profiles/r06d_final_rocprofv3_c3_summary.txt" bash tools/prof.sh r06g_c3 --config c3 > $O/prof_c3.log 2>&1;;
    c5) JSON_ARGS="--json gpurun_out/pmc_derived.json --config c5 --source profiles/r06d_final_rocprofv3_c5_summary.txt" BENCH_CMD="python $PWD/tools/bench_c5.py --no-events --steps 10 --warmup 2" bash tools/prof.sh r06g_c5 > $O/prof_c5.log 2>&1;;
  esac
  cp gpurun_out/prof_r06g_$t/summary.txt $O/rocprofv3_${t}_summary.txt
  rm -rf gpurun_out/prof_r06g_$t/trace gpurun_out/prof_r06g_$t/pmc1 gpurun_out/prof_r06g_$t/pmc2 gpurun_out/prof_r06g_$t/pmc3 gpurun_out/prof_r06g_$t/pmc4
done
cp gpurun_out/pmc_derived.json $O/pmc_derived.json
timeout 200 python bench.py --presteps 2500 --steps 100 --warmup 10 --no-cpu-baseline --no-extras --all-kernels > $O/motion.json 2> $O/motion.err; grep -v "No rigid" $O/motion.err | head -8
timeout 200 python tools/bench_c5.py --no-events --steps 20 --warmup 3 2>/dev/null | tail -1 > $O/bench_c5.json; cat $O/bench_c5.json | cut -c1-300
timeout 200 python bench.py --config c4 --steps 50 --warmup 5 --no-cpu-baseline --no-extras --motion-step 0 2>/dev/null | tail -1 | cut -c1-260 > $O/bench_c4_one_gpu.json; cat $O/bench_c4_one_gpu.json; echo
timeout 200 python tools/slab_size_probe.py --steps 200 > $O/slab_size_probe.json 2>/dev/null; cat $O/slab_size_probe.json
SPH_COMM_TRANSPORT=shm+ipc timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 20 --warmup 5 > $O/torchrun_two_ranks.json 2> $O/torchrun_two_ranks.err; tail -c 600 $O/torchrun_two_ranks.json; echo; tail -3 $O/torchrun_two_ranks.err
# the driver's 8-rank job with all ranks on this one GPU (strong scaling of the 1.23 M scene: the headline form for N > 1 since round 6)
SPH_COMM_TRANSPORT=shm+ipc SPH_COMM_TIMEOUT_S=120 timeout 600 python bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline --motion-step 0 > $O/eight_ranks_one_gpu_bench.json 2> $O/eight_ranks_one_gpu_bench.err; tail -c 900 $O/eight_ranks_one_gpu_bench.json; echo
# C3 with the solvers' own stop tests, from rest and in motion
timeout 300 python bench.py --config c3 --measured-iterations --steps 20 --warmup 5 --no-cpu-baseline --no-extras --motion-step 0 2>/dev/null | tail -1 | cut -c1-400 > $O/bench_c3_measured_rest.json
