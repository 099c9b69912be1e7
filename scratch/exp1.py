import os, sys, subprocess, json
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
what = os.environ["EXP_WHAT"]
if what == "sw":
    b = synth.sw_pairs(65536,150,150,seed=2, related_frac=0.25)
    dev = ctx.sw_batch(b); info = dev.info(); ms = timeit(dev, 20)
    b2 = synth.sw_pairs(262144,32,512,seed=4)
    dev2 = ctx.sw_batch(b2); info2 = dev2.info(); ms2 = timeit(dev2, 3)
    print("SW maxC=%%s lib=%%s | C2: %%.3f ms %%.0f GCUPS eff %%.3f waves %%d | mixed: %%.3f ms %%.0f GCUPS eff %%.3f launches %%d" %% (os.environ.get("AGX_SW_MAX_C"), os.path.basename(agx.LIB_PATH), ms, 65536*22500/ms/1e6, info.cells/info.padded_cells, info.n_waves, ms2, b2.cells(False)/ms2/1e6, info2.cells/info2.padded_cells, info2.n_launches), flush=True)
else:
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
for lib in ("accelerating-genomics_amd/libagx.so", "scratch/libagx_u0.so"):
    for c in (40, 32, 28, 24, 20, 16, 12, 8):
        env = dict(os.environ, EXP_WHAT="sw", AGX_SW_MAX_C=str(c), AGX_LIB_PATH=os.path.join(ROOT, lib))
        subprocess.run([sys.executable, "-c", child], env=env)
for c in (32, 24, 16, 12, 8, 4):
    env = dict(os.environ, EXP_WHAT="ph", AGX_PHMM_MAX_C=str(c))
    subprocess.run([sys.executable, "-c", child], env=env)
