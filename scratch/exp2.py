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
    for _ in range(3):
        ctx.timer_start()
        for _ in range(reps): dev.launch()
        best = min(best, ctx.timer_stop()/reps)
    return best
p3 = synth.phmm_regions(64,64,16,100,300,seed=3)
p5 = synth.phmm_regions(64,32,16,250,500,seed=5)
out = []
for p,name in ((p3,"C3"),(p5,"C5/8")):
    for prec,pn in ((agx.PHMM_F32,"f32"),(agx.PHMM_F64,"f64"),(agx.PHMM_F64_FMA,"fma")):
        dev = ctx.phmm_batch(p, prec); info = dev.info(); ms = timeit(dev, 3)
        out.append("%%s %%s %%.3f ms %%.1f Mp/s eff %%.3f" %% (name, pn, ms, p.n_pairs/ms/1e3, info.cells/info.padded_cells))
        dev.close()
print("PH maxC=%%s | " %% os.environ.get("AGX_PHMM_MAX_C") + " | ".join(out), flush=True)
''' % ROOT
for c in (32, 24, 16, 12, 8):
    env = dict(os.environ, AGX_PHMM_MAX_C=str(c))
    subprocess.run([sys.executable, "-c", child], env=env)
