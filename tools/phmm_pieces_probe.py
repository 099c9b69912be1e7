"""Where the time of agx_phmm_forward on config 5's full batch goes when it is sent through in pieces: per-piece
create / launch call durations on the host, then the results."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
ctx = agx.Context(0)
p = synth.phmm_regions(512, 32, 16, 250, 500, seed=5)
ctx.phmm_forward(p, agx.PHMM_F64)
for K in (1, 2, 4, 8):
    for rep in range(2):
        subs = [p.regions(512 * k // K, 512 * (k + 1) // K) for k in range(K)]
        ctx.sync()
        t0 = time.perf_counter()
        devs, marks = [], []
        for s in subs:
            ta = time.perf_counter(); d = ctx.phmm_batch(s, agx.PHMM_F64); tb = time.perf_counter(); d.launch(); tc = time.perf_counter()
            devs.append(d); marks.append((tb - ta, tc - tb))
        ctx.sync()
        tr = time.perf_counter()
        outs = [d.results(want_sums=False)[0] for d in devs]
        t1 = time.perf_counter()
        for d in devs: d.close()
    print("K=%d total %.2f ms (fills done at %.2f): creates %s launches %s results %.2f ms" % (K, (t1 - t0) * 1e3, (tr - t0) * 1e3, ["%.2f" % (a * 1e3) for a, _ in marks], ["%.2f" % (b * 1e3) for _, b in marks], (t1 - tr) * 1e3), flush=True)
for _ in range(3):
    t0 = time.perf_counter(); l = ctx.phmm_forward(p, agx.PHMM_F64); print("agx_phmm_forward %.2f ms" % ((time.perf_counter() - t0) * 1e3))
