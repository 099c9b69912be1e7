"""Four launches each of config 4's and config 5's per-GPU shards (bench.py's weak legs: 131 072 mixed SW pairs, seed 4;
32 768 PairHMM pairs R=250 H=500 in AGX_PHMM_F64, seed 5) for the FETCH_SIZE / WRITE_SIZE passes that give
profiles/traffic.json its "sw_fill_c4shard" and "phmm_fill_c5shard" figures (tools/prof_traffic_extra.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
ctx = agx.Context(0)
c2 = synth.sw_pairs(65536, 150, 150, seed=2, related_frac=0.25)  # config 2: its pack kernel (grid 2048 workgroups)
d2 = ctx.sw_batch(c2)
d2.launch()
print("c2 checksum", int(d2.scores().sum()), "raw bytes", c2.bases.size)
d2.close()
c4 = synth.sw_pairs((1 << 20) // 8, 32, 512, seed=4)
print("c4 raw bytes", c4.bases.size)
d4 = ctx.sw_batch(c4)
for _ in range(4):
    d4.launch()
ctx.sync()
print("c4 checksum", int(d4.scores().sum()))
d4.close()
c5 = synth.phmm_regions(64, 32, 16, 250, 500, seed=5)
d5 = ctx.phmm_batch(c5, agx.PHMM_F64)
for _ in range(4):
    d5.launch()
ctx.sync()
print("c5 checksum %.6f" % float(d5.results()[0].sum()))
d5.close()
