"""Envelope of read trains for the packed float PairHMM fill (not built): K reads streamed through one lane group without
draining cost K (R + 1) + G - 1 steps instead of K (R + G - 1).  Emulated by batches of the same cells with reads K (R + 1)
- 1 bases long and 1 / K as many of them (same tables, same cells, fewer and longer waves) at config 3's size (65 536
pairs, 8192 waves) and at four times that (262 144 pairs): what a train of K could gain at best, before its bookkeeping."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
ctx = agx.Context(0)
def timeit(dev, reps):
    dev.launch(); ctx.sync(); best = 1e9
    for _ in range(7):
        ctx.timer_start()
        for _ in range(reps): dev.launch()
        best = min(best, ctx.timer_stop() / reps)
    return best
for regions in (64, 256):
    base = None
    for K in (1, 2):  # (synth cuts its reads out of the 300-base haplotypes: longer emulated trains would be clipped)
        reads, R = 64 // K, K * 101 - 1
        p = synth.phmm_regions(regions, reads, 16, R, 300, seed=3)
        dev = ctx.phmm_batch(p, agx.PHMM_F32_FMA); i = dev.info(); ms = timeit(dev, 20 if regions == 64 else 6)
        base = base or ms
        print("%6d C3-pairs' cells, trains of %d emulated (reads of %3d): %5d waves, useful cells %.3f, %.4f ms (%+.1f %% against K = 1)" % (
            regions * 1024, K, R, i.n_waves, i.cells / max(1, i.padded_cells), ms, (base / ms - 1) * 100), flush=True)
        dev.close()
