#!/usr/bin/env python3
"""Wall-clock of the drop-in command lines on full-size inputs: SW on config 4 as a file (1 048 576 pairs,
32..512, 573 MB) and PairHMM on config 5 as a file (262 144 pairs R=250 H=500, 512 batches); output checked by
line count and a checksum of the values against a direct library call."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import accelerating_genomics_amd.api as agx, accelerating_genomics_amd.synth as synth
BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "accelerating-genomics_amd", "bin")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
ctx = agx.Context(0)
with tempfile.TemporaryDirectory(dir="/tmp") as d:
    b = synth.sw_pairs(n, 32, 512, seed=4)
    f = os.path.join(d, "sw.in")
    t0 = time.perf_counter(); synth.write_sw_file(f, b); print("wrote %.0f MB in %.1f s" % (os.path.getsize(f) / 1e6, time.perf_counter() - t0), flush=True)
    want = ctx.sw_score(b)
    for rep in range(7):
        t0 = time.perf_counter()
        o = subprocess.run([os.path.join(BIN, "antidiagonalSmithWaterman"), f], stdout=open(os.path.join(d, "out.txt"), "wb"), stderr=subprocess.PIPE,
                           env=dict(os.environ, AGX_TRACE_CLI="1"))
        dt = time.perf_counter() - t0
        print("SW CLI %d pairs: wall %.3f s; %s" % (n, dt, o.stderr.decode().strip().replace("\n", " | ")), flush=True)
    got = np.array([int(l.split()[1]) for l in open(os.path.join(d, "out.txt"), "rb") if l.startswith(b"Score")], np.int32)
    print("SW CLI output identical to the library call:", bool(np.array_equal(got, want)), flush=True)
    p = synth.phmm_regions(512 * n // (1 << 20) or 1, 32, 16, 250, 500, seed=5)
    f = os.path.join(d, "p.in")
    t0 = time.perf_counter(); synth.write_phmm_file(f, p); print("wrote %.0f MB in %.1f s" % (os.path.getsize(f) / 1e6, time.perf_counter() - t0), flush=True)
    want = ctx.phmm_forward(p, agx.PHMM_F64)
    for chunk in ("65536", "16384"):
        for rep in range(5):
            t0 = time.perf_counter()
            o = subprocess.run([os.path.join(BIN, "antidiagsPairHMM"), f, os.path.join(d, "p.out")], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                               env=dict(os.environ, AGX_CLI_CHUNK_PAIRS=chunk, AGX_TRACE_CLI="1"))
            dt = time.perf_counter() - t0
            print("PairHMM CLI %d pairs (f64, chunks of %s pairs): wall %.3f s rc %d %s" % (p.n_pairs, chunk, dt, o.returncode, o.stderr.decode().strip()[:400]), flush=True)
    got = np.array([float(x) for x in open(os.path.join(d, "p.out")).read().split()])
    print("PairHMM CLI output file == library call to 6 decimals:", bool(got.size == want.size and np.max(np.abs(got - want)) <= 5.1e-7), flush=True)
