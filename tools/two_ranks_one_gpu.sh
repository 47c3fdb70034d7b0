#!/bin/bash
# Two ranks x 1.23 M particles sharing ONE MI355X over the push transport: the per-rank cost of sharding with a live neighbour (dead particles,
# records, waits) that a one-rank run cannot show.  Prints ms/step for the variants given as label:ENV pairs and a rocprofv3 kernel-stats table
# of one rank (profiles/r04_two_ranks_one_gpu.txt was made with this).
#   tools/two_ranks_one_gpu.sh r04x "fused:" "none:SPH_NO_SLAB_PRESEND=1 SPH_NO_SLAB_FUSED_FIELDS=1" "slow:SPH_SLAB_LAYOUT=slow"
set -u
O=gpurun_out/$1; shift; mkdir -p $O
export SPH_COMM_TRANSPORT=shm+ipc
A="--gpus 2 --steps 100 --warmup 10 --no-extras --motion-step 0"
for v in "$@"; do
  l=${v%%:*}; e=${v#*:}
  env $e timeout 120 python bench.py $A > $O/two_$l.json 2> $O/two_$l.err
  python -c "
import json; d=json.loads(open('$O/two_$l.json').read().strip().split('\n')[-1]); print('two ranks one GPU', '$l', '%.4f ms/step' % d['ms_per_step'], d['config']['parallelism'])"
done
R=$(pwd); cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $R/$O/two_rank_trace -o t --output-format csv -- python $R/bench.py --gpus 2 --steps 50 --warmup 10 --no-extras --motion-step 0 > /dev/null 2> $R/$O/two_rank_trace.err; cd $R
find $O/two_rank_trace -name "*.csv" -size +512k -delete
python - "$O" <<'PY'
import csv, re, glob, sys
for f in glob.glob(sys.argv[1] + '/two_rank_trace/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(k_\w+)(<[^>]*>)?', r['Name'])
        print('%-44s calls %5s avg %9.1f us' % ((m.group(0) if m else r['Name'])[:44], r['Calls'], float(r['AverageNs']) / 1e3))
PY
