"""The packed float fill near zero steps: config 3's shape with reads of 1, 5, 10, 25, 50 bases (16 lanes x 19 columns), with trains and
without -- what a launch costs when its waves hardly step (table build, records, haplotype loads, epilogue, launch ramp)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
ctx = agx.Context(0)
for opt, name in ((agx.PHMM_TRAINS_OFF, "plain"), (agx.PHMM_TRAINS_ON, "trains")):
    ctx.set_option(agx.OPT_PHMM_TRAINS, opt)
    for R in (1, 5, 10, 25, 50):
        p = synth.phmm_regions(64, 64, 16, R, 300, seed=3)
        dev = ctx.phmm_batch(p, agx.PHMM_F32_FMA); i = dev.info()
        dev.launch(); ctx.sync(); best = 1e9
        for _ in range(9):
            ctx.timer_start()
            for _ in range(10): dev.launch()
            best = min(best, ctx.timer_stop() / 10)
        print("%-6s R = %3d: %5d waves of %3d steps, %.4f ms" % (name, R, i.n_waves, i.padded_cells // (i.n_waves * 64 * 19 * 2), best), flush=True)
        dev.close()
