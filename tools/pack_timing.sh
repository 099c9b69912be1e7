#!/bin/bash
# usage: tools/pack_timing.sh <tag>   -> gpurun_out/<tag>_pack_timing.log, gpurun_out/<tag>_pack_kernels.csv
set -e
tag=${1:-pack}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o pack -- python3 $GRAFT_REPO_ROOT/tools/pack_timing.py > $GRAFT_REPO_ROOT/gpurun_out/${tag}_pack_timing.log 2>&1
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/${tag}_pack_kernels.csv
t=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
python3 - "$t" >> $GRAFT_REPO_ROOT/gpurun_out/${tag}_pack_timing.log <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "pack" in n or "plan" in n:
        by[(n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60], r["Grid_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (n, g), v in sorted(by.items()):
    v.sort()
    print("%-62s grid %-10s n %3d  median %9.1f us  min %9.1f us" % (n, g, len(v), v[len(v) // 2], v[0]))
P
