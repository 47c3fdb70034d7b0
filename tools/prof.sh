#!/bin/bash
# tools/prof.sh <tag> [bench args...] -- rocprofv3 kernel-trace/stats + PMC passes of bench.py on the GPU box.
# Writes under gpurun_out/prof_<tag>/ ; summarise with tools/prof_summary.py and copy into profiles/.
set -u
TAG=$1; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
# BENCH_CMD="python $REPO/tools/bench_c5.py --no-events" tools/prof.sh <tag>  profiles another driver (C5)
BENCH=${BENCH_CMD:-"python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --motion-step 0 $*"}
[ -n "${SKIP_TRACE:-}" ] || timeout 150 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- $BENCH > $OUT/trace.log 2>&1
i=0
for CTRS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "FETCH_SIZE GRBM_GUI_ACTIVE" \
            "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  if [ -n "${PMC_SETS:-}" ] && [[ " $PMC_SETS " != *" $i "* ]]; then continue; fi
  timeout 150 rocprofv3 --kernel-trace --pmc $CTRS -d $OUT/pmc$i -o pmc --output-format csv -- $BENCH > $OUT/pmc$i.log 2>&1
done
cd $REPO
# JSON_ARGS="--json gpurun_out/pmc_derived.json --config c3 --source profiles/<name>.txt" also merges the derived figures into that file
python tools/prof_summary.py $OUT ${JSON_ARGS:-} > $OUT/summary.txt 2>&1
tail -60 $OUT/summary.txt
