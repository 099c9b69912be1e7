"""agx_sw_batch_create on config 2, config 4's shard and the full 1 048 576 mixed pairs, a few times each, so that
`rocprofv3 --kernel-trace --stats` (tools/pack_timing.sh) shows the pack kernel's durations per shape; scores of the
first two are checked against the oracle (threaded) so a fast but wrong pack cannot pass."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
from tests import oracle_api
orc = oracle_api.load()
ctx = agx.Context(0)
for name, b, check in (("C2 65536x150x150", synth.sw_pairs(65536, 150, 150, seed=2, related_frac=0.25), True),
                       ("C4 shard 131072 mixed", synth.sw_pairs(131072, 32, 512, seed=4), True),
                       ("C4 full 1M mixed", synth.sw_pairs(1 << 20, 32, 512, seed=4), False)):
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); dev = ctx.sw_batch(b); ts.append(time.perf_counter() - t0)
        dev.launch(); s = dev.scores(); dev.close()
    ok = ""
    if check:
        sub = b.subset(np.arange(0, b.n_pairs, 8))
        ok = "scores of every 8th pair == oracle: %s" % bool(np.array_equal(s[::8], oracle_api.sw_batch_mt(orc, sub)))
    print("%-24s create median %.3f ms min %.3f ms; checksum %d %s" % (name, np.median(ts) * 1e3, min(ts) * 1e3, int(s.astype(np.int64).sum()), ok), flush=True)
