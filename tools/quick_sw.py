"""SW kernel timings on the bench shapes (config 2, config 4's per-GPU shard, 262144 and 1 M mixed pairs);
AGX_SW_KERNEL=pk1|i32 selects the other kernels.  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
ctx = agx.Context(0)
def timeit(dev, reps):
    for _ in range(3): dev.launch()
    ctx.sync()
    best = 1e9
    for _ in range(5):
        ctx.timer_start()
        for _ in range(reps): dev.launch()
        best = min(best, ctx.timer_stop() / reps)
    return best
print("kernel:", os.environ.get("AGX_SW_KERNEL", "default"), flush=True)
b = synth.sw_pairs(65536, 150, 150, seed=2, related_frac=0.25)
dev = ctx.sw_batch(b); info = dev.info(); ms = timeit(dev, 100)
print("SW C2: %.4f ms %.0f GCUPS eff %.3f waves %d checksum %d" % (ms, 65536 * 22500 / ms / 1e6, info.cells / info.padded_cells, info.n_waves, int(dev.scores().sum())), flush=True); dev.close()
for n, reps in ((131072, 20), (262144, 10), (1 << 20, 3)):
    if n > 262144 and os.environ.get("QUICK_SMALL"): continue
    b = synth.sw_pairs(n, 32, 512, seed=4)
    dev = ctx.sw_batch(b); info = dev.info(); ms = timeit(dev, reps)
    print("SW mixed %7d: %.3f ms %.0f GCUPS eff %.3f launches %d checksum %d" % (n, ms, b.cells(False) / ms / 1e6, info.cells / info.padded_cells, info.n_launches, int(dev.scores().sum())), flush=True); dev.close()
