"""What a launch of the double PairHMM fill (looked-up priors) spends outside its steps: config 5's shard shape (32 768 pairs against
500-base haplotypes, 16 lanes x 32 columns, 8192 waves) with reads of 60, 125, 250, 500 bases -> per-step time and intercept."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
ctx = agx.Context(0)
for prec, name in ((agx.PHMM_F64, "f64"), (agx.PHMM_F64_FMA, "fma")):
    pts = []
    for R in (60, 125, 250, 500):
        p = synth.phmm_regions(32, 64, 16, R, 500, seed=3)
        dev = ctx.phmm_batch(p, prec); i = dev.info()
        dev.launch(); ctx.sync(); best = 1e9
        for _ in range(7):
            ctx.timer_start()
            for _ in range(6): dev.launch()
            best = min(best, ctx.timer_stop() / 6)
        steps = i.padded_cells // (i.n_waves * 64 * 32)
        pts.append((steps, best))
        print("%s R = %3d: %5d waves of %3d steps, %.4f ms" % (name, R, i.n_waves, steps, best), flush=True)
        dev.close()
    (x0, y0), (x1, y1) = pts[0], pts[-1]
    slope = (y1 - y0) / (x1 - x0)
    print("%s per step %.4f us per launch, intercept %.2f us" % (name, slope * 1e3, (y0 - slope * x0) * 1e3), flush=True)
