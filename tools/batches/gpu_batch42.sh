#!/bin/bash
O=gpurun_out/b42; mkdir -p $O
SPH_COMM_TRANSPORT=shm timeout 900 python bench.py --gpus 4 --steps 20 --warmup 5 --no-cpu-baseline > $O/x4.json 2> $O/x4.err; echo rc=$?
python - <<PY
import json
d=json.loads(open("$O/x4.json").read().strip().splitlines()[-1])
print(d["n_gpus"], d["ms_per_step"], d["value"], d["config"]["parallelism"], d["config"]["particles"], (d.get("in_motion") or {}).get("ms_per_step"))
c=d.get("c4_strong_scaling"); print({k:c[k] for k in ("n_gpus","ms_per_step","value","slab_cuts","owned_per_rank")} if c else None)
PY
tail -3 $O/x4.err | cut -c1-300
