"""agx_sw_batch_create (plan + upload + device pack) and one-shot agx_sw_score on config 2 and on mixed batches;
AGX_TRACE_CREATE=1 prints the stages.  Run on the GPU box."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
ctx = agx.Context(0)
for name, b in (("C2 65536 150x150", synth.sw_pairs(65536, 150, 150, seed=2, related_frac=0.25)),
                ("mixed 131072", synth.sw_pairs(131072, 32, 512, seed=4)), ("mixed 1048576", synth.sw_pairs(1 << 20, 32, 512, seed=4))):
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); dev = ctx.sw_batch(b); ts.append(time.perf_counter() - t0); dev.close()
    one = []
    for _ in range(8):
        t0 = time.perf_counter(); sc = ctx.sw_score(b); one.append(time.perf_counter() - t0)
    print("%s: create median %.3f ms min %.3f ms; one-shot score median %.3f ms min %.3f ms; checksum %d" % (
        name, np.median(ts) * 1e3, min(ts) * 1e3, np.median(one) * 1e3, min(one) * 1e3, int(sc.sum())), flush=True)
