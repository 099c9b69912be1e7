"""Accuracy of the packed float PairHMM fill against the double oracle: max |d log10 L| and max relative error on
log10 L, for the fast and (AGX_PHMM_PLAIN_CELL=1) the plain cell, with and without the GATK prior.  Run on the GPU box."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
from tests import oracle_api
orc = oracle_api.load()
ctx = agx.Context(0)
cases = [("golden 10s", agx.read_phmm_text(os.path.join(ROOT, "tests", "golden", "phmm_10s.in"))[0]),
         ("C3 sample (8 regions)", synth.phmm_regions(8, 64, 16, 100, 300, seed=3)),
         ("C5-shaped (2 regions)", synth.phmm_regions(2, 32, 16, 250, 500, seed=5)),
         ("mixed", synth.phmm_regions(8, 32, 8, 150, 380, seed=7, jitter=100))]
for name, b in cases:
    for flag, variant in ((0, 0), (agx.PHMM_GATK_PRIOR, 3)):
        _, ref = orc.phmm_batch(b, variant)
        for prec, pn in ((agx.PHMM_F32_FMA, "f32fma"), (agx.PHMM_F32, "f32")):
            got = ctx.phmm_forward(b, prec | flag)
            ok = np.isfinite(ref)
            d = np.abs(got[ok] - ref[ok])
            print("%-22s %-5s gatk %d cell %s: max |dlog10| %.3e (1e-6 on L = 4.34e-7), max rel on log10 %.3e" % (
                name, pn, 1 if flag else 0, "plain" if os.environ.get("AGX_PHMM_PLAIN_CELL") else "fast", d.max(), (d / np.abs(ref[ok])).max()), flush=True)
