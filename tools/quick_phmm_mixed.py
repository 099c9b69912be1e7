"""PairHMM on MIXED regions (reads 50-150 x haplotypes 280-380, the shape of real calling regions): packed float and double
fills, cells/s next to the uniform config-3 figure.  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
ctx = agx.Context(0)
def timeit(dev, reps):
    for _ in range(3): dev.launch()
    ctx.sync()
    best = 1e9
    for _ in range(5):
        ctx.timer_start()
        for _ in range(reps): dev.launch()
        best = min(best, ctx.timer_stop() / reps)
    return best
for name, p in (("uniform R=100 H=300", synth.phmm_regions(64, 64, 16, 100, 300, seed=3)),
                ("mixed R 50-150, H 280-380", synth.phmm_regions(64, 64, 16, 150, 380, seed=3, jitter=100)),
                ("mixed, 256 regions", synth.phmm_regions(256, 64, 16, 150, 380, seed=4, jitter=100))):
    for prec in (agx.PHMM_F32_FMA, agx.PHMM_F64):
        dev = ctx.phmm_batch(p, prec); info = dev.info(); ms = timeit(dev, 10)
        print("%s prec %d: %d pairs %.4f ms %.2f Mpairs/s %.2f Tcells/s eff %.3f waves %d launches %d" % (
            name, prec, p.n_pairs, ms, p.n_pairs / ms / 1e3, info.cells / ms / 1e9, info.cells / info.padded_cells, info.n_waves, info.n_launches), flush=True)
        dev.close()
