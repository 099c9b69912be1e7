import sys, time, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
ctx = agx.Context(0)
def timeit(dev, reps):
    dev.launch(); ctx.sync()
    best = 1e9
    for _ in range(5):
        ctx.timer_start()
        for _ in range(reps): dev.launch()
        best = min(best, ctx.timer_stop()/reps)
    return best
b = synth.sw_pairs(65536,150,150,seed=2, related_frac=0.25)
dev = ctx.sw_batch(b); info = dev.info(); ms = timeit(dev, 50)
print("SW C2: %.4f ms %.0f GCUPS eff %.3f waves %d" % (ms, 65536*22500/ms/1e6, info.cells/info.padded_cells, info.n_waves), flush=True); dev.close()
b = synth.sw_pairs(262144,32,512,seed=4)
dev = ctx.sw_batch(b); info = dev.info(); ms = timeit(dev, 5)
print("SW mixed 256k: %.3f ms %.0f GCUPS eff %.3f launches %d" % (ms, b.cells(False)/ms/1e6, info.cells/info.padded_cells, info.n_launches), flush=True); dev.close()
for (nr,reads,haps,R,H,name,reps) in ((64,64,16,100,300,"C3",20),(64,32,16,250,500,"C5/8",8)):
    p = synth.phmm_regions(nr,reads,haps,R,H,seed=3)
    for prec,pn in ((agx.PHMM_F32_FMA,"f32fma"),(agx.PHMM_F32,"f32"),(agx.PHMM_F64,"f64"),(agx.PHMM_F64_FMA,"fma")):
        dev = ctx.phmm_batch(p, prec); info = dev.info(); ms = timeit(dev, reps)
        print("PHMM %s %s: %.4f ms %.2f Mpairs/s eff %.3f waves %d" % (name, pn, ms, p.n_pairs/ms/1e3, info.cells/info.padded_cells, info.n_waves), flush=True); dev.close()
# haplotypes beyond one wave's span: striped kernel, one pair per wavefront
p = synth.phmm_regions(16, 32, 16, 250, 5000, seed=6)
for prec, pn in ((agx.PHMM_F64, "f64"), (agx.PHMM_F64_FMA, "fma")):
    dev = ctx.phmm_batch(p, prec); info = dev.info(); ms = timeit(dev, 2)
    print("PHMM long 250x5000 %s: %.3f ms %.3f Mpairs/s %.1f GCUPS eff %.3f waves %d" % (pn, ms, p.n_pairs/ms/1e3, info.cells/ms/1e6, info.cells/info.padded_cells, info.n_waves), flush=True); dev.close()
# substitution-matrix mode (int32 kernel + LDS lookup) and the int32 kernel itself for comparison
m = agx.SwMatrix.build(synth.AMINO, synth.BLOSUM62, -11, -1)
b = synth.protein_pairs(65536, 150, 150, seed=2, related_frac=0.25)
dev = ctx.sw_batch(b, matrix=m); info = dev.info(); ms = timeit(dev, 20)
print("SW BLOSUM62 65536x~150x150: %.4f ms %.0f GCUPS eff %.3f" % (ms, info.cells/ms/1e6, info.cells/info.padded_cells), flush=True); dev.close()
# mixed shapes (reads 80..150, haplotypes 200..400 within and across regions)
p = synth.phmm_regions(256, 64, 16, 150, 400, seed=8, jitter=0)
parts = [synth.phmm_regions(64, 64, 16, int(R), int(H), seed=80 + k, jitter=int(j)) for k, (R, H, j) in enumerate([(150, 400, 70), (100, 300, 20), (120, 250, 50), (150, 380, 100)])]
for prec, pn in ((agx.PHMM_F32_FMA, "f32fma"), (agx.PHMM_F64, "f64")):
    tot_ms = 0.0; tot_cells = 0; tot_pairs = 0
    for q in parts:
        dev = ctx.phmm_batch(q, prec); info = dev.info(); ms = timeit(dev, 5)
        print("PHMM mixed part %s: %.3f ms %.2f Mpairs/s %.0f GCUPS eff %.3f launches %d" % (pn, ms, q.n_pairs/ms/1e3, info.cells/ms/1e6, info.cells/info.padded_cells, info.n_launches), flush=True); dev.close()
