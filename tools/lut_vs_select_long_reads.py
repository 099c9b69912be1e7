"""Long reads in the double PairHMM fill: the looked-up-prior kernel's 56-byte table rows against the selecting kernel's 33-byte rows
(AGX_PHMM_NO_LUT=1, tuning build) -- beyond 20 KB per wave fewer than two waves per SIMD fit a CU's LDS.  16 384 pairs against
704-base haplotypes, reads of 300 ... 700 bases, kernel-only launch times."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
ctx = agx.Context(0)
tag = "selecting (33-byte rows)     " if "AGX_PHMM_NO_LUT" in os.environ else "looked up, whole tables      " if "AGX_PHMM_NO_RING_FORCE_LUT" in os.environ else "looked up, ring of 256 rows  "
for R in (250, 360, 420, 500, 600, 700):
    p = synth.phmm_regions(16, 64, 16, R, 704, seed=3)
    dev = ctx.phmm_batch(p, agx.PHMM_F64); i = dev.info()
    dev.launch(); ctx.sync(); best = 1e9
    for _ in range(5):
        ctx.timer_start()
        for _ in range(3): dev.launch()
        best = min(best, ctx.timer_stop() / 3)
    print("%s R = %3d: %5d waves, %.4f ms, %.3f ps per padded cell" % (tag, R, i.n_waves, best, best * 1e9 / i.padded_cells), flush=True)
    dev.close()
