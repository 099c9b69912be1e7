#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into one markdown summary.
usage: summarize_prof.py gpurun_out/prof_<tag>"""
import collections
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
print("# rocprofv3 summary (%s)\n" % os.path.basename(d))


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    return name.split("(")[0][:60]


for f in glob.glob(os.path.join(d, "kt", "**", "*kernel_stats.csv"), recursive=True):
    print("## kernel stats (`rocprofv3 --kernel-trace --stats`, bench.py --steps 500 --warmup 50)\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for r in csv.DictReader(open(f)):
        print("| %s | %s | %.3f | %.2f | %.2f | %.2f | %s |" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
              float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
    print()
for f in glob.glob(os.path.join(d, "kt", "**", "*kernel_trace.csv"), recursive=True):
    # the same trace split by grid size: bench.py launches one kernel class for several workloads
    # (config 2's 2048-wave launch and config 4's classes share sw_fill_pk<38>), --stats lumps them
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        by[(short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))].append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("## kernel durations by launch shape (from the kernel trace; headline launches: sw_fill_pk2<38> with 512 "
          "workgroups = 2048 waves, phmm_fill_pk<19, true, true, true> -- config 3 with read trains -- with 4096; the int32 leg is sw_fill_i32d<38> with 512)\n")
    print("| kernel | workgroups | calls | avg us | min us | max us |\n|---|---|---|---|---|---|")
    for (k, wg), v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        if k.startswith("__amd"):
            continue
        print("| %s | %d | %d | %.2f | %.2f | %.2f |" % (k, wg, len(v), sum(v) / len(v), min(v), max(v)))
    print()
    res = {}
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        res[k] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size"),
                  r.get("Workgroup_Size_X"), r.get("Grid_Size_X"))
    # rocprofv3's VGPR_Count is a granulated figure (it showed 100 for sw_fill_pk2<38, 4>, whose code object says 193):
    # the registers below are the code objects' own .vgpr_count / .agpr_count / .sgpr_count (tools/kernel_resources.py)
    try:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from kernel_resources import kernel_resources
        code = kernel_resources()
        fixed = {}
        for k, v in res.items():
            name = k.replace("void ", "")
            hit = code.get(name) or next((code[c] for c in code if c[:60] == name), None)
            fixed[k] = ((hit["vgpr"], hit["agpr"], hit["sgpr"]) + tuple(v[3:])) if hit else v
        res = fixed
    except Exception as e:  # the library is not there: the trace's own columns, with this warning
        print("(code objects not readable here -- %s: VGPR / AGPR / SGPR are rocprofv3's granulated columns)\n" % e)
    print("## per-dispatch resources (registers from the code objects, the rest from the kernel trace)\n")
    print("(LDS B is the static allocation only; the PairHMM kernels take their read tables as dynamic LDS: "
          "32 B x (steps + G - 1) rows per table in the packed kernel -- 4160 B per one-wave workgroup on config 3 -- "
          "and 33 B per row in the double kernel)\n")
    print("| kernel | VGPR | AGPR | SGPR | LDS B | scratch | wg | grid |\n|---|---|---|---|---|---|---|---|")
    for k, v in res.items():
        print("| %s | %s |" % (k, " | ".join(str(x) for x in v)))
    print()
try:
    print("bench line under the kernel-trace run: `%s`\n" % open(os.path.join(d, "kt_bench.json")).read().strip()[:400])
except OSError:
    pass

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
if agg:
    print("## PMC counters (separate `--pmc` passes, mean per dispatch)\n")
    names = sorted({c for k in agg for c in agg[k]})
    print("| kernel | " + " | ".join(names) + " |")
    print("|---|" + "---|" * len(names))
    traffic = {}
    for k in agg:
        row = []
        for c in names:
            v = agg[k].get(c)
            row.append("%.4g" % (sum(v) / len(v)) if v else "")
        print("| %s | %s |" % (k, " | ".join(row)))
        fs, ws = agg[k].get("FETCH_SIZE"), agg[k].get("WRITE_SIZE")
        if fs and ws:
            # MI355X_MICROARCH.md: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the
            # bytes of wide coalesced reads -> doubled (upper bound for narrow reads); WRITE_SIZE exact.
            traffic[k] = {"fetch_kib_raw": sum(fs) / len(fs), "write_kib": sum(ws) / len(ws),
                          "hbm_bytes_corrected": (2 * sum(fs) / len(fs) + sum(ws) / len(ws)) * 1024}
    print()
    if traffic:
        print("## HBM traffic per launch (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, KiB -> bytes)\n")
        for k, v in traffic.items():
            print("- %s: fetch raw %.1f KiB, write %.1f KiB -> %.3f MB corrected" % (k, v["fetch_kib_raw"], v["write_kib"], v["hbm_bytes_corrected"] / 1e6))
        json.dump(traffic, open(os.path.join(d, "traffic_raw.json"), "w"), indent=1)
