"""What a config-3 launch of the packed float fill spends outside its steps: 65 536 pairs against 300-base haplotypes (16 lanes x 19
columns, 8192 waves without read trains) with reads of 50, 100, 200, 300 bases -> per-step time and the intercept (table build,
the sixteen last-row blocks, epilogue, launch ramp); then the same with trains (4096 waves of two reads)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
ctx = agx.Context(0)
for opt, name in ((agx.PHMM_TRAINS_OFF, "plain"), (agx.PHMM_TRAINS_ON, "trains")):
    ctx.set_option(agx.OPT_PHMM_TRAINS, opt)
    pts = []
    for R in (50, 100, 200, 300):
        p = synth.phmm_regions(64, 64, 16, R, 300, seed=3)
        dev = ctx.phmm_batch(p, agx.PHMM_F32_FMA); i = dev.info()
        dev.launch(); ctx.sync(); best = 1e9
        for _ in range(9):
            ctx.timer_start()
            for _ in range(10): dev.launch()
            best = min(best, ctx.timer_stop() / 10)
        steps = i.padded_cells // (i.n_waves * 64 * 19 * 2)
        pts.append((steps, best))
        print("%-6s R = %3d: %5d waves of %3d steps, %.4f ms" % (name, R, i.n_waves, steps, best), flush=True)
        dev.close()
    (x0, y0), (x1, y1) = pts[0], pts[-1]
    slope = (y1 - y0) / (x1 - x0)
    print("%-6s per step %.4f us per launch, intercept %.2f us" % (name, slope * 1e3, (y0 - slope * x0) * 1e3), flush=True)
