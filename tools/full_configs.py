#!/usr/bin/env python3
"""Time BASELINE configs 4 and 5 at full size on ONE GPU (the 8-GPU runs shard them; numbers for DESIGN.md)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
ctx = agx.Context(0)
def timeit(dev, reps):
    dev.launch(); ctx.sync()
    best = 1e9
    for _ in range(3):
        ctx.timer_start()
        for _ in range(reps): dev.launch()
        best = min(best, ctx.timer_stop() / reps)
    return best
b = synth.sw_pairs(1 << 20, 32, 512, seed=4)
t0 = time.perf_counter(); dev = ctx.sw_batch(b); tc = time.perf_counter() - t0
i = dev.info(); ms = timeit(dev, 3)
print("C4 full: 1048576 SW pairs U[32,512]: %.2f ms/launch, %.0f GCUPS (sentinel excluded), useful cells %.3f, %d launches, %d waves; create %.0f ms"
      % (ms, b.cells(False) / ms / 1e6, i.cells / i.padded_cells, i.n_launches, i.n_waves, tc * 1e3), flush=True)
dev.close()
p = synth.phmm_regions(512, 32, 16, 250, 500, seed=5)
for prec, name in ((agx.PHMM_F64, "F64"), (agx.PHMM_F64_FMA, "F64_FMA"), (agx.PHMM_F32, "F32")):
    t0 = time.perf_counter(); dev = ctx.phmm_batch(p, prec); tc = time.perf_counter() - t0
    i = dev.info(); ms = timeit(dev, 3)
    print("C5 full: 262144 PairHMM pairs R=250 H=500 %s: %.2f ms/launch, %.2f M pairs/s, %.0f GCUPS, useful cells %.3f; create %.0f ms"
          % (name, ms, p.n_pairs / ms / 1e3, p.cells() / ms / 1e6, i.cells / i.padded_cells, tc * 1e3), flush=True)
    dev.close()
