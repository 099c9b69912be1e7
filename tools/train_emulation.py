import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
ctx = agx.Context(0)
def timeit(dev, reps):
    dev.launch(); ctx.sync(); best=1e9
    for _ in range(7):
        ctx.timer_start()
        for _ in range(reps): dev.launch()
        best=min(best, ctx.timer_stop()/reps)
    return best
for (reads, R, name) in ((64, 100, "K=1 (C3)"), (32, 202, "K=2 emu"), (16, 406, "K=4 emu"), (8, 814, "K=8 emu")):
    p = synth.phmm_regions(64, reads, 16, R, 300, seed=3)
    dev = ctx.phmm_batch(p, agx.PHMM_F32_FMA); i = dev.info(); ms = timeit(dev, 20)
    print("%s: pairs %d waves %d steps~%d  %.4f ms  -> per C3-equivalent %.4f ms" % (name, p.n_pairs, i.n_waves, R + 15, ms, ms), flush=True); dev.close()
