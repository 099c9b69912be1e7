import sys, time, numpy as np
sys.path.insert(0, ".")
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
from tests import oracle_api
orc = oracle_api.load()
ctx = agx.Context(0)
for (lo,hi,n) in [(1,70,500),(150,150,512),(32,512,600),(500,999,32)]:
    b = synth.sw_pairs(n, lo, hi, seed=3, related_frac=0.5)
    got = ctx.sw_score(b); ref = orc.sw_batch(b)
    print(lo,hi,n,"match" if np.array_equal(got,ref) else ("MISMATCH %d" % (got!=ref).sum()), flush=True)
b = synth.sw_pairs(65536,150,150,seed=2)
dev = ctx.sw_batch(b); info = dev.info()
print("waves",info.n_waves,"launches",info.n_launches,"eff",info.cells/info.padded_cells)
for rep in range(3):
    dev.launch(); ctx.sync()
    ctx.timer_start()
    for _ in range(10): dev.launch()
    ms = ctx.timer_stop()/10
    print("150x150 64k: %.3f ms  %.1f GCUPS (nominal 150x150)" % (ms, 65536*22500/ms/1e6), flush=True)
b = synth.sw_pairs(262144,32,512,seed=4)
dev = ctx.sw_batch(b); info = dev.info()
print("mixed waves",info.n_waves,"launches",info.n_launches,"eff",info.cells/info.padded_cells)
dev.launch(); ctx.sync(); ctx.timer_start()
for _ in range(5): dev.launch()
ms = ctx.timer_stop()/5
print("mixed 256k: %.3f ms  %.1f GCUPS" % (ms, b.cells(False)/ms/1e6), flush=True)
