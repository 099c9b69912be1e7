"""One agx_sw_score call on 1 048 576 mixed pairs (page-locked source) under rocprofv3 --kernel-trace --memory-copy-trace:
tools/timeline_sw_score.sh prints the device timeline of the last call (copies and kernels, ms from its first event)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
ctx = agx.Context(0)
b = synth.sw_pairs(1 << 20, 32, 512, seed=4)
pinned = "pinned" in sys.argv[1:]
if "second-context" in sys.argv[1:]:  # the process-wide context of agx_*_multi on the same device, as bench.py has it
    agx.sw_score_multi(synth.sw_pairs(4096, 32, 512, seed=1), 1)
if pinned:
    bases = agx.host_array(b.bases.size, np.uint8); bases[:] = b.bases
    off = agx.host_array(b.off.size, np.uint64); off[:] = b.off
    ln = agx.host_array(b.len.size, np.uint32); ln[:] = b.len
    b = synth.SWBatch(bases, off, ln)
for k in range(4):
    time.sleep(0.05)  # a gap in the trace before every call
    t0 = time.perf_counter(); s = ctx.sw_score(b); dt = time.perf_counter() - t0
    print("call %d: %.2f ms, checksum %d" % (k, dt * 1e3, int(s.astype(np.int64).sum())), flush=True)
