"""Mixed-length SW batches (both lengths U[32,512]) at several sizes: one launch for all classes (default) against
one launch per class with class consolidation (AGX_SW_ONE_LAUNCH=0), tail-regime term on/off.  Run on the GPU box;
each configuration in its own process (the knobs are read once)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, %r)
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
out = []
for n, reps in ((2048, 50), (8192, 50), (16384, 30), (65536, 20), (131072, 10), (262144, 6), (1 << 20, 3)):
    b = synth.sw_pairs(n, 32, 512, seed=4)
    dev = ctx.sw_batch(b); i = dev.info(); ms = timeit(dev, reps)
    out.append("%%7d: %%6.0f GCUPS %%.3f ms eff %%.3f L%%d" %% (n, b.cells(False) / ms / 1e6, ms, i.cells / i.padded_cells, i.n_launches))
    dev.close()
print(os.environ.get("LABEL", ""), " | ".join(out), flush=True)
''' % ROOT
for label, env in (("one launch, all classes        ", {"AGX_SW_ONE_LAUNCH": "1"}),
                   ("one launch, tail term off      ", {"AGX_SW_ONE_LAUNCH": "1", "AGX_SW_TAIL_BETA": "0"}),
                   ("one launch, <= 6 classes       ", {"AGX_SW_ONE_LAUNCH": "1", "AGX_SW_MAX_CLASSES": "6"}),
                   ("per-class launches (round 1)   ", {"AGX_SW_ONE_LAUNCH": "0", "AGX_SW_MAX_CLASSES": "6"})):
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, LABEL=label, **env))
