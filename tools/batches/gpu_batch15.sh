#!/bin/bash
O=gpurun_out/b15; mkdir -p $O
SPH_BENCH_FORCE_SLAB=1 SPH_COMM_TRANSPORT=rccl NCCL_DEBUG=WARN timeout 300 python bench.py --no-cpu-baseline --steps 20 --motion-step 0 > $O/a.json 2> $O/a.err; echo "rc=$?"
tail -c 1500 $O/a.err; tail -c 600 $O/a.json
SPH_BENCH_FORCE_SLAB=1 SPH_COMM_TRANSPORT=rccl NCCL_DEBUG=WARN timeout 300 python bench.py --no-cpu-baseline --steps 20 > $O/b.json 2> $O/b.err; echo "rc=$?"
tail -c 1500 $O/b.err; tail -c 300 $O/b.json
dmesg 2>/dev/null | tail -5
