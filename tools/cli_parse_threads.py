#!/usr/bin/env python3
"""The SW command line on config 4 as a 573 MB file, with 1, 2, 4, 8 and all threads reading and scanning the chunks
(AGX_CLI_PARSE_THREADS).  usage: cli_parse_threads.py [pairs] [thread counts, comma separated].  Run on the GPU box."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import accelerating_genomics_amd.synth as synth
BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "accelerating-genomics_amd", "bin")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
with tempfile.TemporaryDirectory(dir="/tmp") as d:
    b = synth.sw_pairs(n, 32, 512, seed=4)
    f = os.path.join(d, "sw.in")
    synth.write_sw_file(f, b)
    for thr in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("1", "2", "4", "8", "0", "1", "4", "0")):
        best = None
        for rep in range(3):
            t0 = time.perf_counter()
            o = subprocess.run([os.path.join(BIN, "antidiagonalSmithWaterman"), f], stdout=open(os.path.join(d, "out.txt"), "wb"), stderr=subprocess.PIPE,
                               env=dict(os.environ, AGX_TRACE_CLI="1", AGX_CLI_PARSE_THREADS=thr))
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]: best = (dt, o.stderr.decode().strip().replace("\n", " | "))
        print("parse threads %s: wall %.3f s; %s" % (thr, best[0], best[1]), flush=True)
