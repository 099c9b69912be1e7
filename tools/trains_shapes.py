"""Read trains on the wide classes of the packed float PairHMM fill: kernel-only launch times of uniform batches whose tilings are
16 x 19 (config 3), 16 x 30, 15 x 30, 16 x 32 (config 5's shard in packed float) with trains (AGX_OPT_PHMM_TRAINS on) and without
(off), same process, same box."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
ctx = agx.Context(0)
def timeit(dev, reps):
    dev.launch(); ctx.sync(); best = 1e9
    for _ in range(7):
        ctx.timer_start()
        for _ in range(reps): dev.launch()
        best = min(best, ctx.timer_stop() / reps)
    return best
for name, args, reps in (("R=100 H=200 65536 pairs", (64, 64, 16, 100, 200), 20), ("R=100 H=300 65536 pairs", (64, 64, 16, 100, 300), 20), ("R=100 H=350 65536 pairs", (64, 64, 16, 100, 350), 16), ("R=100 H=400 65536 pairs", (64, 64, 16, 100, 400), 16), ("R=150 H=420 65536 pairs", (64, 64, 16, 150, 420), 12), ("R=100 H=480 65536 pairs", (64, 64, 16, 100, 480), 12),
                         ("R=100 H=450 65536 pairs", (64, 64, 16, 100, 450), 12), ("R=250 H=500 32768 pairs", (32, 64, 16, 250, 500), 8),
                         ("R=250 H=500 131072 pairs", (128, 64, 16, 250, 500), 3), ("R=150 H=250 131072 pairs", (128, 64, 16, 150, 250), 6)):
    p = synth.phmm_regions(*args, seed=3)
    out = []
    for opt in (agx.PHMM_TRAINS_OFF, agx.PHMM_TRAINS_ON):
        ctx.set_option(agx.OPT_PHMM_TRAINS, opt)
        dev = ctx.phmm_batch(p, agx.PHMM_F32_FMA); i = dev.info(); ms = timeit(dev, reps)
        out.append((i.n_waves, i.cells / max(1, i.padded_cells), ms)); dev.close()
    print("%-26s plain %5d waves useful %.3f %.4f ms | trains %5d waves useful %.3f %.4f ms (%+.1f %%)" % (
        name, out[0][0], out[0][1], out[0][2], out[1][0], out[1][1], out[1][2], (out[0][2] / out[1][2] - 1) * 100), flush=True)
