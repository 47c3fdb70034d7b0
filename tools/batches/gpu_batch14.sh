#!/bin/bash
# single-rank slab overhead, 2-rank shm runs (C2 weak, all three solvers), step overhead
O=gpurun_out/b14; mkdir -p $O
python bench.py --no-cpu-baseline > $O/c2.json 2> $O/c2.err
SPH_BENCH_FORCE_SLAB=1 python bench.py --no-cpu-baseline > $O/c2_slab1.json 2> $O/c2_slab1.err
SPH_BENCH_FORCE_SLAB=1 SPH_COMM_TRANSPORT=rccl python bench.py --no-cpu-baseline > $O/c2_slab1_rccl.json 2> $O/c2_slab1_rccl.err
SPH_COMM_TRANSPORT=shm python bench.py --gpus 2 --no-cpu-baseline --steps 50 > $O/c2_x2.json 2> $O/c2_x2.err
SPH_COMM_TRANSPORT=shm python bench.py --gpus 2 --no-cpu-baseline --steps 30 --config c3 --no-c4 > $O/c3_x2.json 2> $O/c3_x2.err
SPH_COMM_TRANSPORT=shm python bench.py --gpus 2 --no-cpu-baseline --steps 30 --method pcisph --no-c4 > $O/pci_x2.json 2> $O/pci_x2.err
python tools/step_overhead.py > $O/step_overhead.txt 2>&1
for f in c2 c2_slab1 c2_slab1_rccl c2_x2 c3_x2 pci_x2; do echo "== $f"; python - <<PY
import json
try:
    d=json.loads(open("$O/$f.json").read().strip().splitlines()[-1])
    print(d["n_gpus"], d["ms_per_step"], d["value"], d["config"].get("parallelism"), d.get("in_motion",{}).get("ms_per_step"), d.get("c4_strong_scaling"))
except Exception as e:
    print("ERR", e); print(open("$O/$f.err").read()[-800:])
PY
done
tail -12 $O/step_overhead.txt
