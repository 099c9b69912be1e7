"""Envelope of read trains for the packed float PairHMM fill (not built): K reads streamed through one lane group without
draining cost K (R + 1) + G - 1 steps instead of K (R + G - 1).  Emulated by batches of the same cells with reads K (R + 1)
- 1 bases long and 1 / K as many of them (same tables, same cells, fewer and longer waves) at config 3's size (65 536
pairs, 8192 waves) and at four times that (262 144 pairs): what a train of K could gain at best, before its bookkeeping.
Reads are random bases here (a fill's duration does not depend on its values; only launches are timed, no results taken),
so that trains longer than the 300-base haplotypes can be emulated."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
ctx = agx.Context(0)
def timeit(dev, reps):
    dev.launch(); ctx.sync(); best = 1e9
    for _ in range(7):
        ctx.timer_start()
        for _ in range(reps): dev.launch()
        best = min(best, ctx.timer_stop() / reps)
    return best
def batch(regions, reads, haps, R, H, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(regions):
        hs = [synth._ACGT[rng.integers(0, 4, size=H)].tobytes() for _ in range(haps)]
        q = lambda lo, hi: (rng.integers(lo, hi, size=R) + 33).astype(np.uint8).tobytes()
        rs = [(synth._ACGT[rng.integers(0, 4, size=R)].tobytes(), q(6, 42), q(39, 46), q(39, 46), bytes([43]) * R) for _ in range(reads)]
        out.append((rs, hs))
    return synth.phmm_from_regions(out)
for regions in (64, 256):
    base = None
    for K in (1, 2, 4, 8):
        reads, R = 64 // K, K * 101 - 1
        p = batch(regions, reads, 16, R, 300, seed=3)
        dev = ctx.phmm_batch(p, agx.PHMM_F32_FMA); i = dev.info(); ms = timeit(dev, 20 if regions == 64 else 6)
        base = base or ms
        print("%6d C3-pairs' cells, trains of %d emulated (reads of %3d): %5d waves, useful cells %.3f, %.4f ms (%+.1f %% against K = 1)" % (
            regions * 1024, K, R, i.n_waves, i.cells / max(1, i.padded_cells), ms, (base / ms - 1) * 100), flush=True)
        dev.close()
