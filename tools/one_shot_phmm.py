"""agx_phmm_forward (host buffers in, log10 likelihoods out: plan + H2D + fill + D2H) on config 3 in packed float and on config 5's
shard in double; tools/one_shot_phmm.sh repeats it with AGX_PHMM_NO_TRAINS=1 / AGX_PHMM_NO_ROWS=1 (tuning build)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
ctx = agx.Context(0)
for name, p, prec in (("config 3 packed float", synth.phmm_regions(64, 64, 16, 100, 300, seed=3), agx.PHMM_F32_FMA),
                      ("config 5 shard double", synth.phmm_regions(32, 64, 16, 250, 500, seed=5), agx.PHMM_F64)):
    ctx.phmm_forward(p, prec); ctx.phmm_forward(p, prec)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); ctx.phmm_forward(p, prec); ts.append(time.perf_counter() - t0)
    cr = []
    for _ in range(10):
        t0 = time.perf_counter(); d = ctx.phmm_batch(p, prec); cr.append(time.perf_counter() - t0); d.close()
    print("%-24s %-40s one-shot min %.3f median %.3f ms; create alone min %.3f median %.3f ms" % (
        name, " ".join(k + "=" + os.environ[k] for k in ("AGX_PHMM_NO_TRAINS", "AGX_PHMM_NO_ROWS") if k in os.environ) or "default",
        1e3 * min(ts), 1e3 * float(np.median(ts)), 1e3 * min(cr), 1e3 * float(np.median(cr))), flush=True)
