#!/bin/bash
O=gpurun_out/b35; mkdir -p $O
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/out.txt 2> $O/err.txt ) 2>&1 | grep real
wc -l $O/out.txt; python - <<PY
import json
d=json.loads(open("$O/out.txt").read().strip())
for k in ("metric","value","unit","n_gpus","steps","warmup","ms_per_step","higher_is_better","scaling","vs_baseline","dtype","data"): print(k, d[k])
print(d["config"]["workload"], d["roofline"], sep="\n"); print({k:v for k,v in d["cpu_baseline"].items() if k not in ("threads_sweep",)})
PY
