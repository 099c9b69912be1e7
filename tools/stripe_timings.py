"""Striped PairHMM kernel: time per columns-per-lane setting (AGX_PHMM_STRIPE_C is read once per process)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys; sys.path.insert(0, %r)
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
ctx = agx.Context(0)
for (R, H) in ((250, 5000), (100, 2100), (1000, 4000)):
    p = synth.phmm_regions(16, 32, 16, R, H, seed=6)
    for prec, pn in ((agx.PHMM_F64, "f64"), (agx.PHMM_F64_FMA, "fma")):
        dev = ctx.phmm_batch(p, prec); info = dev.info()
        dev.launch(); ctx.sync(); best = 1e9
        for _ in range(3):
            ctx.timer_start(); dev.launch(); best = min(best, ctx.timer_stop())
        print("  %%dx%%d %%s: %%.3f ms %%.1f GCUPS (padded %%.1f) eff %%.3f" %% (R, H, pn, best, info.cells/best/1e6, info.padded_cells/best/1e6, info.cells/info.padded_cells), flush=True)
        dev.close()
''' % ROOT
for c in (24, 26, 28, 30):
    print("AGX_PHMM_STRIPE_C=%d" % c, flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, AGX_PHMM_STRIPE_C=str(c)), check=True)
