"""Mixed SW batches (32-512) across batch sizes: what the planner's tail-regime rule buys (AGX_SW_TAIL_BETA=0 switches
it off).  usage: sw_tail_rule_check.py [sizes, comma separated].  Run on the GPU box."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
ctx = agx.Context(0)
def timeit(dev, reps):
    dev.launch(); ctx.sync(); best=1e9
    for _ in range(5):
        ctx.timer_start()
        for _ in range(reps): dev.launch()
        best=min(best, ctx.timer_stop()/reps)
    return best
SIZES = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2048, 8192, 16384, 32768, 65536, 98304, 131072, 163840, 196608, 262144, 1048576]
for n in SIZES:
    b = synth.sw_pairs(n, 32, 512, seed=4)
    dev = ctx.sw_batch(b); i = dev.info(); ms = timeit(dev, 3)
    print("  n=%d: %.3f ms %.0f GCUPS eff %.3f launches %d waves %d" % (n, ms, b.cells(False)/ms/1e6, i.cells/i.padded_cells, i.n_launches, i.n_waves), flush=True); dev.close()
if len(sys.argv) <= 1:
    b = synth.sw_pairs(131072, 100, 300, seed=9)
    dev = ctx.sw_batch(b); i = dev.info(); ms = timeit(dev, 3)
    print("  U[100,300] n=131072: %.3f ms %.0f GCUPS waves %d" % (ms, b.cells(False)/ms/1e6, i.n_waves))
