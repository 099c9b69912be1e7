#!/bin/bash
# PMC passes over tools/prof_phmm_lut.py (each in its own run, no trace domains): gpurun_out/prof_lut_<tag>/
TAG=${1:-r02}; OUT=gpurun_out/prof_lut_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "VALUBusy VALUUtilization OccupancyPercent" "LDSBankConflict MemUnitBusy"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -- python3 tools/prof_phmm_lut.py > $OUT/pmc$i.out 2> $OUT/pmc$i.err || { echo "pmc pass $i failed"; tail -3 $OUT/pmc$i.err; }
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if "phmm_fill" not in k: continue
    print(k)
    for c, v in sorted(cs.items()): print("  %-24s %.4g (mean of %d dispatches)" % (c, sum(v) / len(v), len(v)))
PY
