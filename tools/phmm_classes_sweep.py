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
parts = [synth.phmm_regions(64, 64, 16, int(R), int(H), seed=80 + k, jitter=int(j)) for k, (R, H, j) in enumerate([(150, 400, 70), (100, 300, 20), (120, 250, 50), (150, 380, 100)])]
big = synth.phmm_regions(256, 64, 16, 150, 380, seed=90, jitter=100)
for prec, pn in ((agx.PHMM_F32_FMA, "pkf"), (agx.PHMM_F64, "f64")):
    out=[]
    for q in parts + [big]:
        dev = ctx.phmm_batch(q, prec); i = dev.info(); ms = timeit(dev, 3)
        out.append("%.0f(%d)" % (i.cells/ms/1e6, i.n_launches)); dev.close()
    print("  %s GCUPS(launches): %s" % (pn, "  ".join(out)), flush=True)
'''
for k in (1, 2, 3, 4, 6):
    print("AGX_PHMM_MAX_CLASSES=%d (cap; rule 1 per 8192 waves)" % k, flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, AGX_PHMM_MAX_CLASSES=str(k)))
