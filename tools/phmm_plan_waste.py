"""Where the padded cells of a PairHMM plan go (plan-only batches, no GPU needed): the tuning build's AGX_TRACE_CREATE line of
the planner for the corpus leg (tests/golden/phmm_10s.in x 19) and configs 3 and 5, double and packed float."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("AGX_TRACE_CREATE", "1")
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "phmm_10s.in")
corpus = synth.phmm_repeat(synth.parse_phmm_text(open(gold, "rb").read()), 19)
for name, b in (("corpus_10s", corpus), ("config 3", synth.phmm_regions(64, 64, 16, 100, 300, seed=3)), ("config 5 shard", synth.phmm_regions(32, 64, 16, 250, 500, seed=5))):
    for prec, pn in ((agx.PHMM_F64, "double"), (agx.PHMM_F32_FMA, "packed float")):
        print("== %s, %s" % (name, pn), flush=True); sys.stderr.flush()
        d = agx.PhmmBatchDev(None, b, prec); i = d.info()
        print("   waves %d, useful %.4f" % (i.n_waves, i.cells / i.padded_cells), flush=True)
        d.close()
