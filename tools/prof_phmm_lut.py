"""A few launches of config 5's shard in AGX_PHMM_F64 (phmm_fill_lut_w2<32>; AGX_PHMM_NO_LUT=1: phmm_fill_w2<double, 32>) for
rocprofv3 --pmc passes: `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python3 tools/prof_phmm_lut.py`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
ctx = agx.Context(0)
p = synth.phmm_regions(64, 32, 16, 250, 500, seed=5)
dev = ctx.phmm_batch(p, agx.PHMM_F64)
for _ in range(4):
    dev.launch()
ctx.sync()
l, _ = dev.results()
print("checksum %.6f" % float(l.sum()))
dev.close()
