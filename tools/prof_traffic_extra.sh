#!/bin/bash
# FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes over tools/prof_traffic_extra.py -> per-launch HBM bytes
# (KiB -> bytes, FETCH_SIZE doubled: gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md) of the two fills.
TAG=${1:-r02}; OUT=gpurun_out/prof_traffic_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for set in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $set --output-format csv -d $OUT/$set -- python3 tools/prof_traffic_extra.py > $OUT/$set.out 2> $OUT/$set.err || { echo "$set pass failed"; tail -3 $OUT/$set.err; }
done
python3 - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in acc.items():
    if not any(t in k for t in ("sw_fill", "phmm_fill", "sw_pack", "sw_plan", "DeviceRadixSort", "DeviceScan", "radix_sort", "scan")): continue
    f = sum(cs["FETCH_SIZE"]) / max(1, len(cs["FETCH_SIZE"])); w = sum(cs["WRITE_SIZE"]) / max(1, len(cs["WRITE_SIZE"])); n = len(cs["FETCH_SIZE"])
    name = k.replace("(anonymous namespace)::", "")
    out[(name[:name.index("(")] if "(" in name else name)[-90:]] = {"fetch_kib_raw": f, "write_kib": w, "hbm_bytes_corrected": (2 * f + w) * 1024, "dispatches": n}
print(json.dumps(out, indent=1))
PY
