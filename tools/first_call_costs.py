"""First-use costs in a fresh process: what the hipvers window (launch -> scores, one shot per process) pays."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
t0 = time.perf_counter(); ctx = agx.Context(0); print("ctx create %.1f ms" % ((time.perf_counter() - t0) * 1e3))
b = synth.sw_pairs(25000, 64, 64, seed=64)
t0 = time.perf_counter(); dev = ctx.sw_batch(b); print("batch create %.1f ms" % ((time.perf_counter() - t0) * 1e3))
out = np.empty(b.n_pairs, np.int32)
for k in range(3):
    t0 = time.perf_counter(); dev.launch(); t1 = time.perf_counter(); ctx.sync(); t2 = time.perf_counter(); dev.scores(out); t3 = time.perf_counter()
    print("round %d: launch call %.3f ms, sync %.3f ms, scores %.3f ms" % (k, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
