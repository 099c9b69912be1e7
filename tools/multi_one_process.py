"""bench.py's multi_one_process leg alone: agx_sw_score_multi / agx_phmm_forward_multi on the full configs 4 and 5 through
N devices of one process (default: all visible), host buffers in, results out; best of 5 and median after a warm call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else agx.device_count()
c4 = synth.sw_pairs(1 << 20, 32, 512, seed=4)
c5 = synth.phmm_regions(512, 32, 16, 250, 500, seed=5)
def best(fn, reps=5):
    fn(); fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, float(np.median(ts)) * 1e3, r
a, am, s = best(lambda: agx.sw_score_multi(c4, n))
print("agx_sw_score_multi(%d) config 4: min %.2f ms median %.2f ms (%.0f GCUPS host-inclusive), checksum %d" % (n, a, am, c4.cells(False) / a / 1e6, int(s.astype(np.int64).sum())), flush=True)
for prec, name in ((agx.PHMM_F64, "f64"), (agx.PHMM_F32_FMA, "f32fma")):
    a, am, l = best(lambda: agx.phmm_forward_multi(c5, prec, n))
    print("agx_phmm_forward_multi(%d) config 5 %s: min %.2f ms median %.2f ms (%.2f M pairs/s host-inclusive), checksum %.6f" % (n, name, a, am, c5.n_pairs / a / 1e3, float(l.sum())), flush=True)
