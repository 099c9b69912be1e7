"""Host-inclusive one-shot timings (agx_sw_score / agx_phmm_forward: host buffers in, results out) on the
bench shapes, pageable and pinned sources, plus the create-time breakdown (AGX_TRACE_CREATE, tuning build)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
ctx = agx.Context(0)
def pinned_copy(b):
    bases = agx.host_array(b.bases.size, np.uint8); bases[:] = b.bases
    off = agx.host_array(b.off.size, np.uint64); off[:] = b.off
    ln = agx.host_array(b.len.size, np.uint32); ln[:] = b.len
    return synth.SWBatch(bases, off, ln)
for name, b, reps in (("C2 65536x150x150", synth.sw_pairs(65536, 150, 150, seed=2, related_frac=0.25), 20),
                      ("C4 shard 131072 mixed", synth.sw_pairs(131072, 32, 512, seed=4), 5),
                      ("C4 full 1M mixed", synth.sw_pairs(1 << 20, 32, 512, seed=4), 3)):
    for kind, bb in (("pageable", b), ("pinned", pinned_copy(b))):
        ctx.sw_score(bb)  # warm the pools
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); s = ctx.sw_score(bb); ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); dev = ctx.sw_batch(bb); tc = time.perf_counter() - t0
        dev.close()
        print("SW %-22s %-8s one-shot median %.3f ms min %.3f ms (%.0f GCUPS host-inclusive); create alone %.3f ms; checksum %d" % (
            name, kind, 1e3 * float(np.median(ts)), 1e3 * min(ts), b.cells(False) / min(ts) / 1e9, 1e3 * tc, int(s.sum())), flush=True)
for name, p, prec, reps in (("C3 65536 R100 H300 f32fma", synth.phmm_regions(64, 64, 16, 100, 300, seed=3), agx.PHMM_F32_FMA, 10),
                            ("C5 shard 32768 R250 H500 f64", synth.phmm_regions(64, 32, 16, 250, 500, seed=5), agx.PHMM_F64, 5),
                            ("C5 full 262144 R250 H500 f64", synth.phmm_regions(512, 32, 16, 250, 500, seed=5), agx.PHMM_F64, 3)):
    ctx.phmm_forward(p, prec)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); l = ctx.phmm_forward(p, prec); ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); dev = ctx.phmm_batch(p, prec); tc = time.perf_counter() - t0
    dev.close()
    print("PHMM %-30s one-shot median %.3f ms min %.3f ms (%.2f M pairs/s host-inclusive); create alone %.3f ms; checksum %.6f" % (
        name, 1e3 * float(np.median(ts)), 1e3 * min(ts), p.n_pairs / min(ts) / 1e6, 1e3 * tc, float(l.sum())), flush=True)
