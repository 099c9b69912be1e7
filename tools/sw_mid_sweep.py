"""Mid-size mixed SW batch (config 4's per-GPU shard: 131072 pairs of 32..512): class count x widest class."""
import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys; sys.path.insert(0, %r)
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
ctx = agx.Context(0)
def timeit(dev, reps):
    dev.launch(); ctx.sync(); best=1e9
    for _ in range(5):
        ctx.timer_start()
        for _ in range(reps): dev.launch()
        best=min(best, ctx.timer_stop()/reps)
    return best
b = synth.sw_pairs(131072, 32, 512, seed=4)
dev = ctx.sw_batch(b); i = dev.info(); ms = timeit(dev, 5)
print("%%.3f ms %%.0f GCUPS eff %%.3f launches %%d waves %%d" %% (ms, b.cells(False)/ms/1e6, i.cells/i.padded_cells, i.n_launches, i.n_waves), flush=True)
''' % ROOT
for mc in (40, 30, 24, 20):
    for k in (1, 2, 3, 4, 6):
        print("MAX_C=%d classes<=%d: " % (mc, k), end="", flush=True)
        subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, AGX_SW_MAX_C=str(mc), AGX_SW_MAX_CLASSES=str(k), AGX_SW_WAVES_PER_CLASS="1"))
