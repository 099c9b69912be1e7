import sys, os, subprocess
CHILD=r'''
import sys, os; sys.path.insert(0, "/root/repo")
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
ctx = agx.Context(0)
def timeit(dev, reps):
    dev.launch(); ctx.sync(); best=1e9
    for _ in range(5):
        ctx.timer_start()
        for _ in range(reps): dev.launch()
        best=min(best, ctx.timer_stop()/reps)
    return best
b = synth.sw_pairs(65536,150,150,seed=2, related_frac=0.25)
dev = ctx.sw_batch(b); i = dev.info(); ms = timeit(dev, 50)
print("C=%s: %.4f ms %.0f GCUPS eff %.3f waves %d" % (os.environ.get("AGX_SW_FORCE_C","auto"), ms, 65536*22500/ms/1e6, i.cells/i.padded_cells, i.n_waves), flush=True)
p = synth.phmm_regions(64,64,16,100,300,seed=3)
dev = ctx.phmm_batch(p, agx.PHMM_F32_FMA); i = dev.info(); ms = timeit(dev, 20)
print("   PHMM C3 pkf C=%s: %.4f ms %.2f Mpairs/s eff %.3f waves %d" % (os.environ.get("AGX_PHMM_FORCE_C","auto"), ms, p.n_pairs/ms/1e3, i.cells/i.padded_cells, i.n_waves), flush=True)
'''
subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ))
for C, PC in ((16, 15), (20, 17), (22, 19), (26, 20), (30, 22), (32, 25), (38, 28), (40, 30)):
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, AGX_SW_FORCE_C=str(C), AGX_PHMM_FORCE_C=str(PC)))
