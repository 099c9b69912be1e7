"""What a config-2 launch spends outside its steps: 65 536 pairs with shorter sides of 150 and longer sides of 150, 300, 600, 1200
(same tiling: 4 lanes x 38 columns, 2048 waves = one residency), kernel-only launch times -> per-step time and the
intercept (prologue, epilogue, launch ramp)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
ctx = agx.Context(0)
rng = np.random.default_rng(1)
pts = []
for ly in (150, 300, 600, 1200):
    seqs = []
    for _ in range(65536):
        seqs.append(synth._ACGT[rng.integers(0, 4, size=150)].tobytes() + b"\n")
        seqs.append(synth._ACGT[rng.integers(0, 4, size=ly)].tobytes() + b"\n")
    b = synth.sw_from_seqs(seqs)
    dev = ctx.sw_batch(b); i = dev.info()
    dev.launch(); ctx.sync(); best = 1e9
    for _ in range(9):
        ctx.timer_start()
        for _ in range(10): dev.launch()
        best = min(best, ctx.timer_stop() / 10)
    pts.append((ly, best))
    print("150 x %4d: %d waves, %.4f ms, %.0f GCUPS" % (ly, i.n_waves, best, 65536 * 151 * (ly + 1) / best / 1e6), flush=True)
    dev.close()
(x0, y0), (x1, y1) = pts[0], pts[-1]
slope = (y1 - y0) / (x1 - x0)
print("per step %.4f us, intercept %.2f us (at 0 steps)" % (slope * 1e3, (y0 - slope * (x0 + 4)) * 1e3))
