import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import sys, os, numpy as np
sys.path.insert(0, %r)
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
print("SW C2 forceC=%%s: %%.4f ms %%.0f GCUPS eff %%.3f waves %%d" %% (os.environ.get("AGX_SW_FORCE_C"), ms, 65536*22500/ms/1e6, info.cells/info.padded_cells, info.n_waves), flush=True)
''' % ROOT
for c in (0, 36, 38, 40, 38, 40):
    env = dict(os.environ)
    if c: env["AGX_SW_FORCE_C"] = str(c)
    subprocess.run([sys.executable, "-c", child], env=env)
