"""Config 2 and a mixed batch in the reference's window (launch -> scores in page-locked host memory), per-step wall
clock.  (Round 2 tried letting the fill store its scores in page-locked host memory itself instead of the D2H copy:
0.2117 vs 0.2130 ms on config 2, 1.369 vs 1.317 ms on the mixed batch -- scattered 4-byte PCIe writes; dropped.  Polling
hipStreamQuery instead of hipStreamSynchronize: no difference, 0.1945 ms either way -- the runtime's wait already spins.)
Run on the GPU box."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
ctx = agx.Context(0)
for name, b in (("C2", synth.sw_pairs(65536, 150, 150, seed=2, related_frac=0.25)), ("mixed 131072", synth.sw_pairs(131072, 32, 512, seed=4))):
    dev = ctx.sw_batch(b)
    out = agx.host_array(b.n_pairs, np.int32)
    for _ in range(50): dev.launch(); dev.scores(out)
    ts = []
    for _ in range(300):
        t0 = time.perf_counter(); dev.launch(); dev.scores(out); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e3
    print("%s: window median %.4f ms min %.4f ms checksum %d" % (name, np.median(ts), ts.min(), int(out.sum())), flush=True)
    dev.close()

p = synth.phmm_regions(64, 64, 16, 100, 300, seed=3)
dev = ctx.phmm_batch(p, agx.PHMM_F32_FMA)
out = agx.host_array(p.n_pairs, np.float64)
for _ in range(50): dev.launch(); dev.results((out, None), want_sums=False)
ts = []
for _ in range(300):
    t0 = time.perf_counter(); dev.launch(); dev.results((out, None), want_sums=False); ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
print("PairHMM C3 f32fma: window median %.4f ms min %.4f ms checksum %.6f" % (np.median(ts), ts.min(), float(out.sum())), flush=True)
