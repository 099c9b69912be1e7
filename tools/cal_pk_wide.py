import sys, os, subprocess
CHILD=r'''
import sys, os; sys.path.insert(0, "/root/repo")
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
ctx = agx.Context(0)
def timeit(dev, reps):
    dev.launch(); ctx.sync(); best=1e9
    for _ in range(3):
        ctx.timer_start()
        for _ in range(reps): dev.launch()
        best=min(best, ctx.timer_stop()/reps)
    return best
C=int(os.environ["CAL_C"])
p = synth.phmm_regions(192, 32, 16, 128, 8*C, seed=2)
dev = ctx.phmm_batch(p, agx.PHMM_F32_FMA); i = dev.info(); ms = timeit(dev, 3)
print("ph pkf C=%2d waves %6d padded %.3e ms %.3f  ps/padded-cell %.3f" % (C, i.n_waves, i.padded_cells, ms, ms*1e9/i.padded_cells), flush=True); dev.close()
'''
for C in (28, 29, 30, 31, 32):
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, CAL_C=str(C), AGX_PHMM_FORCE_C=str(C)))
