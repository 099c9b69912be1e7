#!/bin/bash
# The round's evidence in ONE gpurun: rocprofv3 kernel trace + PMC passes over bench.py (tools/profile_bench.sh), the
# FETCH_SIZE / WRITE_SIZE passes over config 4's and config 5's shards and the pack / plan kernels (tools/prof_traffic_extra.sh),
# profiles/traffic.json rebuilt from THOSE passes and stamped with the commit in profiles/.commit, then the bench line
# that quotes it.   usage: tools/profile_round.sh <tag>      (outputs under gpurun_out/, copied to profiles/ by the caller)
set -o pipefail
TAG=${1:-r03z}
export TMPDIR=/tmp
bash tools/profile_bench.sh $TAG > gpurun_out/${TAG}_profile_bench.log 2>&1 || { echo "profile_bench failed"; tail -5 gpurun_out/${TAG}_profile_bench.log; exit 1; }
bash tools/prof_traffic_extra.sh $TAG > gpurun_out/${TAG}_traffic_extra.json 2> gpurun_out/${TAG}_traffic_extra.err || { echo "traffic passes failed"; exit 1; }
python3 - $TAG <<'P'
import json, os, sys
tag = sys.argv[1]
raw = json.load(open("gpurun_out/prof_%s/traffic_raw.json" % tag))
extra = json.load(open("gpurun_out/%s_traffic_extra.json" % tag))
def pick(table, *needles):
    for k, v in table.items():
        if all(n in k for n in needles):
            return int(round(v["hbm_bytes_corrected"]))
    return None
commit = open("profiles/.commit").read().strip() if os.path.exists("profiles/.commit") else "unknown"
out = {"sw_fill": pick(raw, "sw_fill_pk2<38"), "phmm_fill": pick(raw, "phmm_fill_pk<19, true, true, true") or pick(raw, "phmm_fill_pk_w3<19"), "sw_fill_int32": pick(raw, "sw_fill_i32d") or pick(raw, "sw_fill<"),
       "sw_fill_c4shard": pick(extra, "sw_fill_pk2_any"), "phmm_fill_c5shard": pick(extra, "phmm_fill_lut_w2<32"),
       "sw_pack_dna_c2_and_c4shard_mean": pick(extra, "sw_pack_dna"),
       "_commit": commit,
       "_source": "gpurun_out/prof_%s (copied to profiles/%s_bench_rocprofv3_summary.md): rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over "
                  "`bench.py --no-extra-configs` (headline launches) and over tools/prof_traffic_extra.py (config 4's / config 5's shards, pack and planning kernels), "
                  "KiB -> bytes, FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); collected in the same gpurun as the bench line "
                  "profiles/%s_bench_steps20.json, code of commit %s" % (tag, tag, tag, commit)}
json.dump(out, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
P
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_steps20.json 2> gpurun_out/${TAG}_bench_steps20.err; echo "bench rc=$?"
cp profiles/traffic.json gpurun_out/${TAG}_traffic.json
cp gpurun_out/prof_$TAG/summary.md gpurun_out/${TAG}_bench_rocprofv3_summary.md
