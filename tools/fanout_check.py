"""Fan-out over side streams vs one stream, for batches that spread over many classes (experiment)."""
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
for n in (16384, 65536, 262144, 1048576):
    b = synth.sw_pairs(n, 32, 512, seed=4)
    dev = ctx.sw_batch(b); i = dev.info(); ms = timeit(dev, 3)
    print("  SW n=%%d: %%.3f ms %%.0f GCUPS launches %%d" %% (n, ms, b.cells(False)/ms/1e6, i.n_launches), flush=True); dev.close()
q = synth.phmm_regions(64, 64, 16, 150, 380, seed=83, jitter=100)
for prec in (agx.PHMM_F32_FMA, agx.PHMM_F64):
    dev = ctx.phmm_batch(q, prec); i = dev.info(); ms = timeit(dev, 3)
    print("  PHMM mixed prec %%d: %%.3f ms %%.0f GCUPS launches %%d" %% (prec, ms, i.cells/ms/1e6, i.n_launches), flush=True); dev.close()
''' % ROOT
for fan in ("1", "0"):
    for mc, wpc in ((19, 1), (6, 2048)):
        print("AGX_FANOUT=%s MAX_CLASSES=%d WAVES_PER_CLASS=%d" % (fan, mc, wpc), flush=True)
        subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, AGX_FANOUT=fan, AGX_SW_MAX_CLASSES=str(mc), AGX_SW_WAVES_PER_CLASS=str(wpc), AGX_PHMM_MAX_CLASSES=str(mc)))
