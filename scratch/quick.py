import sys, time, numpy as np
sys.path.insert(0, ".")
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
ctx = agx.Context(0)
def timeit(dev, reps):
    dev.launch(); ctx.sync()
    ctx.timer_start()
    for _ in range(reps): dev.launch()
    return ctx.timer_stop()/reps
b = synth.sw_pairs(65536,150,150,seed=2)
dev = ctx.sw_batch(b); info = dev.info()
print("SW C2 waves",info.n_waves,"launches",info.n_launches,"eff",info.cells/info.padded_cells, flush=True)
for rep in range(2):
    ms = timeit(dev, 10)
    print("SW 150x150 64k: %.3f ms  %.1f GCUPS (nominal 150x150)" % (ms, 65536*22500/ms/1e6), flush=True)
dev.close()
b = synth.sw_pairs(262144,32,512,seed=4)
dev = ctx.sw_batch(b); info = dev.info()
print("SW mixed waves",info.n_waves,"launches",info.n_launches,"eff",info.cells/info.padded_cells, flush=True)
ms = timeit(dev, 5)
print("SW mixed 256k: %.3f ms  %.1f GCUPS" % (ms, b.cells(False)/ms/1e6), flush=True)
dev.close()
p = synth.phmm_regions(64,64,16,100,300,seed=3)
for prec,name in ((agx.PHMM_F32,"f32"),(agx.PHMM_F64,"f64"),(agx.PHMM_F64_FMA,"f64fma")):
    dev = ctx.phmm_batch(p, prec); info = dev.info()
    ms = timeit(dev, 5)
    print("PHMM C3 %s: waves %d launches %d eff %.3f  %.3f ms  %.3f Mpairs/s  %.1f GCUPS" % (name, info.n_waves, info.n_launches, info.cells/info.padded_cells, ms, p.n_pairs/ms/1e3, p.cells()/ms/1e6), flush=True)
    dev.close()
p = synth.phmm_regions(64,32,16,250,500,seed=5)
for prec,name in ((agx.PHMM_F64,"f64"),(agx.PHMM_F64_FMA,"f64fma"),(agx.PHMM_F32,"f32")):
    dev = ctx.phmm_batch(p, prec); info = dev.info()
    ms = timeit(dev, 3)
    print("PHMM C5/8 %s: waves %d launches %d eff %.3f  %.3f ms  %.3f Mpairs/s  %.1f GCUPS" % (name, info.n_waves, info.n_launches, info.cells/info.padded_cells, ms, p.n_pairs/ms/1e3, p.cells()/ms/1e6), flush=True)
    dev.close()
