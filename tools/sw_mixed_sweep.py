import sys, os, subprocess
CHILD=r'''
import sys; sys.path.insert(0, "/root/repo")
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
ctx = agx.Context(0)
def timeit(dev, reps):
    dev.launch(); ctx.sync(); best=1e9
    for _ in range(5):
        ctx.timer_start()
        for _ in range(reps): dev.launch()
        best=min(best, ctx.timer_stop()/reps)
    return best
for n in (32768, 65536, 131072, 262144, 1048576):
    b = synth.sw_pairs(n, 32, 512, seed=4)
    dev = ctx.sw_batch(b); i = dev.info(); ms = timeit(dev, 3)
    print("  n=%d: %.3f ms %.0f GCUPS eff %.3f launches %d waves %d" % (n, ms, b.cells(False)/ms/1e6, i.cells/i.padded_cells, i.n_launches, i.n_waves), flush=True); dev.close()
'''
for mc, wpc in ((6, 4096), (6, 2048), (6, 1024), (10, 1024), (10, 512)):
    print("MAX_CLASSES=%d WAVES_PER_CLASS=%d" % (mc, wpc), flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, AGX_SW_MAX_CLASSES=str(mc), AGX_SW_WAVES_PER_CLASS=str(wpc)))
