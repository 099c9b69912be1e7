"""Per-class lane time of the packed SW kernels (AGX_SW_KERNEL=pk1 for the first formulation):
one subprocess per width because AGX_SW_FORCE_C is read once.  Run on the GPU box."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, %r)
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
ctx = agx.Context(0)
C = int(os.environ["CAL_C"])
b = synth.sw_pairs(262144, 8*C, 8*C, seed=1, newline=False)
b.len[1::2] = 8*C
dev = ctx.sw_batch(b); i = dev.info()
dev.launch(); ctx.sync()
best = 1e9
for _ in range(3):
    ctx.timer_start()
    for _ in range(3): dev.launch()
    best = min(best, ctx.timer_stop()/3)
print("sw %%s C=%%2d waves %%6d padded %%.3e ms %%.3f  ps/padded-cell %%.4f" %% (os.environ.get("AGX_SW_KERNEL","pk2"), C, i.n_waves, i.padded_cells, best, best*1e9/i.padded_cells), flush=True)
''' % ROOT
for C in range(4, 42, 2):
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, CAL_C=str(C), AGX_SW_FORCE_C=str(C)))
