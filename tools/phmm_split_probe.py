"""Would config 3 (65 536 uniform pairs = 8 waves per SIMD of 16 lanes x 19 columns) gain from a split plan -- most pairs on the
widest tiling (10 lanes x 30 columns: 7 % less lane time per pair when the chip is evenly filled) and the rest on 16 x 19 so that
every SIMD gets whole waves?  Launch time of config-3-shaped batches of `regions` regions (1024 pairs each) under whatever
AGX_PHMM_FORCE_C the tuning build was given: tools/phmm_split_probe.sh adds the parts up."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
ctx = agx.Context(0)
for regions in [int(a) for a in sys.argv[1:]]:
    p = synth.phmm_regions(regions, 64, 16, 100, 300, seed=3)
    dev = ctx.phmm_batch(p, agx.PHMM_F32_FMA); i = dev.info()
    dev.launch(); ctx.sync(); best = 1e9
    for _ in range(9):
        ctx.timer_start()
        for _ in range(20): dev.launch()
        best = min(best, ctx.timer_stop() / 20)
    print("FORCE_C=%-3s %6d pairs: %5d waves, useful cells %.3f, %.4f ms" % (os.environ.get("AGX_PHMM_FORCE_C", "-"), regions * 1024, i.n_waves,
          i.cells / max(1, i.padded_cells), best), flush=True)
    dev.close()
