#!/usr/bin/env python3
"""Wall-clock of the drop-in SW command line on a config-4-sized input (1 048 576 pairs, 32..512)."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import accelerating_genomics_amd.synth as synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
b = synth.sw_pairs(n, 32, 512, seed=4)
with tempfile.TemporaryDirectory(dir="/tmp") as d:
    f = os.path.join(d, "sw.in")
    t0 = time.perf_counter(); synth.write_sw_file(f, b); print("wrote %.0f MB in %.1f s" % (os.path.getsize(f) / 1e6, time.perf_counter() - t0), flush=True)
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "accelerating-genomics_amd", "bin", "antidiagonalSmithWaterman")
    for rep in range(2):
        t0 = time.perf_counter()
        o = subprocess.run([exe, f], stdout=open(os.path.join(d, "out.txt"), "wb"), stderr=subprocess.PIPE, env=dict(os.environ, AGX_TRACE_CREATE="1", AGX_TRACE_CLI="1"))
        dt = time.perf_counter() - t0
        print("CLI %d pairs: wall %.2f s; stderr: %s" % (n, dt, o.stderr.decode().strip().replace("\n", " | ")), flush=True)
    print(subprocess.run(["tail", "-2", os.path.join(d, "out.txt")], capture_output=True).stdout.decode())
