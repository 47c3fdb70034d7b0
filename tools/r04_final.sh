# the round's closing run: GPU suite, the driver's bench command, rocprofv3 trace + PMC passes, C2 in motion per kernel, a rank's share of C4 / C2
O=gpurun_out/r04_final; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/gpu_suite.txt 2>&1; tail -4 $O/gpu_suite.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_driver_line.json 2> $O/bench_driver_line.err; tail -c 200 $O/bench_driver_line.json; echo
rm -rf gpurun_out/prof_r04; bash tools/prof.sh r04 > $O/prof.log 2>&1; tail -3 $O/prof.log; cp gpurun_out/prof_r04/summary.txt $O/rocprofv3_c2_summary.txt
find gpurun_out/prof_r04 -name "*.csv" -size +1M -delete; find gpurun_out/prof_r04 -name "*.db" -delete
timeout 200 python bench.py --presteps 2500 --steps 100 --warmup 10 --no-cpu-baseline --no-extras --all-kernels > $O/motion.json 2> $O/motion.err; grep -v "No rigid" $O/motion.err | head -8
timeout 200 python tools/slab_size_probe.py --steps 200 > $O/slab_size_probe.json 2>/dev/null; cat $O/slab_size_probe.json
# the driver's N > 1 command line (torch.distributed.run exports RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), two ranks sharing this box's one GPU
SPH_COMM_TRANSPORT=shm+ipc timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 20 --warmup 5 > $O/torchrun_two_ranks.json 2> $O/torchrun_two_ranks.err; tail -c 600 $O/torchrun_two_ranks.json; echo; tail -3 $O/torchrun_two_ranks.err
