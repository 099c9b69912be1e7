"""The PairHMM command line on a config-5-shaped file of 1 048 576 pairs (99 MB): wall clock against AGX_CLI_CHUNK_PAIRS and
AGX_PHMM_PRECISION.  Run on the GPU box."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import accelerating_genomics_amd.synth as synth
BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "accelerating-genomics_amd", "bin")
with tempfile.TemporaryDirectory(dir="/tmp") as d:
    f = os.path.join(d, "p.in")
    synth.write_phmm_file(f, synth.phmm_regions(2048, 32, 16, 250, 500, seed=5))
    for prec in ("f64", "f32fma"):
        for chunk in ("16384", "65536", "131072", "262144", "524288"):
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                o = subprocess.run([os.path.join(BIN, "antidiagsPairHMM"), f, os.path.join(d, "p.out")], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                                   env=dict(os.environ, AGX_CLI_CHUNK_PAIRS=chunk, AGX_PHMM_PRECISION=prec, AGX_TRACE_CLI="1"))
                ts.append(time.perf_counter() - t0)
            print("PairHMM CLI 1048576 pairs %-6s chunks of %6s pairs: wall min %.3f median %.3f max %.3f s rc %d | last run: %s" % (
                prec, chunk, min(ts), sorted(ts)[1], max(ts), o.returncode, o.stderr.decode().strip()[-330:]), flush=True)
