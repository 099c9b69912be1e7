#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + stats, then PMC passes (each in its own run,
# never combined with trace domains -- see the gpurun rules).  Output under gpurun_out/prof_<tag>/.
# usage: tools/profile_bench.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 500 --warmup 50 --no-cpu-baseline $*"   # the default bench run, so durations compare with BENCH json
echo "== kernel trace" 
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py $ARGS > $OUT/kt_bench.json 2> $OUT/kt.err || { echo "kernel-trace run failed"; tail -5 $OUT/kt.err; exit 1; }
PMCARGS="--steps 4 --warmup 1 --warm-seconds 0.01 --no-cpu-baseline --no-extra-configs $*"   # headline legs only, a handful of dispatches
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" "VALUBusy VALUUtilization OccupancyPercent" "LDSBankConflict MemUnitBusy" ; do
  i=$((i+1))
  echo "== pmc pass $i: $set"
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -- python3 bench.py $PMCARGS > $OUT/pmc${i}_bench.json 2> $OUT/pmc$i.err || { echo "pmc pass $i failed"; tail -5 $OUT/pmc$i.err; }
done
python3 tools/summarize_prof.py $OUT > $OUT/summary.md
cat $OUT/summary.md
