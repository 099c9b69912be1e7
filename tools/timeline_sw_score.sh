#!/bin/bash
# usage: tools/timeline_sw_score.sh <tag> [pinned]
tag=${1:-tl}; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl_$tag -o tl -- python3 $GRAFT_REPO_ROOT/tools/timeline_sw_score.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/${tag}_timeline.log 2>&1
python3 - /tmp/tl_$tag >> $GRAFT_REPO_ROOT/gpurun_out/${tag}_timeline.log <<'P'
import csv, glob, sys
ev = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + n.split("(")[0][-48:]))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C %s %s B" % (r.get("Direction", ""), r.get("Bytes", r.get("Size", "")))))
ev.sort()
# the last burst: events after the last gap of more than 20 ms
start = 0
for i in range(1, len(ev)):
    if ev[i][0] - max(e[1] for e in ev[:i][-50:]) > 20e6:
        start = i
t0 = ev[start][0]
print("---- device timeline of the last call (ms from its first event): start, end, duration, what")
for a, b, n in ev[start:]:
    print("%8.3f %8.3f %7.3f  %s" % ((a - t0) / 1e6, (b - t0) / 1e6, (b - a) / 1e6, n))
P
