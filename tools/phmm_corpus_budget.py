"""The reference's corpus shape (tests/golden/phmm_10s.in, its regions x 19) against the LDS budget a wave's read tables may
take (AGX_PHMM_TAB_BUDGET, tuning build; default 20 KB = 160 KB / 8 waves per CU): more tables per wave fill the waves
better (useful cells) and leave fewer waves on a CU.  Prints useful cells and the kernel-only launch time; run once per
budget (tools/phmm_corpus_budget.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
corpus = synth.phmm_repeat(synth.parse_phmm_text(open(os.path.join(ROOT, "tests", "golden", "phmm_10s.in"), "rb").read()), 19)
ctx = agx.Context(0)
ref = None
for prec, name in ((agx.PHMM_F64, "f64"), (agx.PHMM_F32_FMA, "f32_fma")):
    dev = ctx.phmm_batch(corpus, prec)
    i = dev.info()
    for _ in range(30):
        dev.launch()
    ctx.sync()
    best = 1e9
    for _ in range(5):
        ctx.timer_start()
        for _ in range(20):
            dev.launch()
        best = min(best, ctx.timer_stop() / 20)
    l, _ = dev.results()
    dev.close()
    print("budget %-8s %-8s waves %5d useful cells %.4f launch %.4f ms -> %.1f M pairs/s, %.2f T cells/s, checksum %.6f" % (
        os.environ.get("AGX_PHMM_TAB_BUDGET", "default"), name, i.n_waves, i.cells / i.padded_cells, best, corpus.n_pairs / best / 1e3, corpus.cells() / best / 1e9, float(l.sum())), flush=True)
