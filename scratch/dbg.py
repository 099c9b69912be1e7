import sys, numpy as np
sys.path.insert(0, ".")
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
from tests import oracle_api
orc = oracle_api.load()
ctx = agx.Context(0)
sc = (12, -100, -50, -7)
b = synth.sw_pairs(1500, 1, 400, seed=abs(sum(sc)) + 50, related_frac=0.6)
dev = ctx.sw_batch(b, sc); dev.launch(); got = dev.scores(); ref = orc.sw_batch_scored(b, sc)
bad = np.nonzero(got != ref)[0]
print(len(bad), "bad of", b.n_pairs)
for p in bad[:8]:
    print(p, "len", b.len[2*p], b.len[2*p+1], "got", got[p], "ref", ref[p])
for sc2 in [(12,-100,-3,-1),(12,-1,-50,-7),(1,-1,-50,-7),(3,-100,-3,-1),(12,-20,-3,-1),(12,-4,-3,-1)]:
    dev = ctx.sw_batch(b, sc2); dev.launch(); g2 = dev.scores(); r2 = orc.sw_batch_scored(b, sc2)
    print(sc2, (g2 != r2).sum())
