#!/usr/bin/env python3
"""Measure, per kernel class (columns per lane), the time one lane spends per padded cell.
The scheduler's tiling choice weighs classes by these numbers (agx_sw.cpp / agx_phmm.cpp).
Run on the GPU box:  python tools/calibrate_classes.py > gpurun_out/calibration.log"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
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
what, C = os.environ["CAL_WHAT"], int(os.environ["CAL_C"])
if what == "sw":
    b = synth.sw_pairs(262144, 8*C, 8*C, seed=1, newline=False)
    b.len[1::2] = 8*C  # both 8C: G = 8 exactly
    dev = ctx.sw_batch(b); i = dev.info(); ms = timeit(dev, 3)
    print("sw C=%%2d waves %%6d padded %%.3e ms %%.3f  ps/padded-cell %%.3f  GCUPS(real) %%.0f" %% (C, i.n_waves, i.padded_cells, ms, ms*1e9/i.padded_cells, i.cells/ms/1e6), flush=True)
else:
    p = synth.phmm_regions(192, 32, 16, 128, 8*C, seed=2)
    for prec, name in ((agx.PHMM_F32, "f32"), (agx.PHMM_F64, "f64"), (agx.PHMM_F64_FMA, "fma"), (agx.PHMM_F32_FMA, "pkf")):
        if os.environ.get("CAL_ONLY_PKF") and name != "pkf": continue
        if C > 32 and prec != agx.PHMM_F32: continue
        if C > 30 and prec == agx.PHMM_F32_FMA: continue
        dev = ctx.phmm_batch(p, prec); i = dev.info(); ms = timeit(dev, 3)
        print("ph %%s C=%%2d waves %%6d padded %%.3e ms %%.3f  ps/padded-cell %%.3f  Mpairs/s %%.1f" %% (name, C, i.n_waves, i.padded_cells, ms, ms*1e9/i.padded_cells, p.n_pairs/ms/1e3), flush=True)
        dev.close()
''' % ROOT
EVEN = tuple(range(4, 42, 2))
for kern in (() if os.environ.get("CAL_ONLY_PH") or os.environ.get("CAL_ONLY_PKF") else ("pk", "i32")):
    print("# SW kernel", kern, flush=True)
    for C in EVEN:
        subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, CAL_WHAT="sw", CAL_C=str(C), AGX_SW_FORCE_C=str(C), AGX_SW_KERNEL=kern))
if os.environ.get("CAL_ONLY_PKF"):  # the packed float kernel has every width 4..30
    for C in range(4, 31):
        subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, CAL_WHAT="ph", CAL_C=str(C), AGX_PHMM_FORCE_C=str(C)))
    sys.exit(0)
for C in EVEN:
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, CAL_WHAT="ph", CAL_C=str(C), AGX_PHMM_FORCE_C=str(C)))
