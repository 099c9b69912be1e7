#!/usr/bin/env python3
"""Reproduce the shape of the reference's only published benchmark (hiprun.sh,
50 000 random sequences = 25 000 alignments, LEN x LEN, block in {32..1024}, mean elapsed with a 90 %
CI, hipvers timing window) with the drop-in `hipvers` on this box.  Writes a markdown table.
    python tools/hipvers_sweep.py [--runs 10] > gpurun_out/hipvers_sweep.md"""
import argparse
import os
import statistics
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import accelerating_genomics_amd.synth as synth  # noqa: E402

MI210_BEST_MS = {64: 4.04, 128: 8.79, 256: 28.30, 512: 110.10, 1024: 649.40}


def run(exe, inp, out, block, warm):
    env = dict(os.environ)
    if warm:
        env["AGX_HIPVERS_WARMUP"] = "1"
    r = subprocess.run([exe, inp, out, str(block)], capture_output=True, env=env, check=True)
    return float([l for l in r.stdout.splitlines() if l.startswith(b"elapsed")][0].split()[1]) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=10)
    ap.add_argument("--alignments", type=int, default=25000)
    args = ap.parse_args()
    exe = os.path.join(ROOT, "accelerating-genomics_amd", "bin", "hipvers")
    print("# hipvers sweep on this box (%d alignments per run, %d runs per cell, ms, mean +- 90%% CI)\n" % (args.alignments, args.runs))
    print("`cold` = the reference's protocol (one launch per process, first-launch cost inside the window); "
          "`warm` = AGX_HIPVERS_WARMUP=1 (one untimed launch first). <block_size> does not shape our launch; "
          "the column is kept for the table's shape.\n")
    print("| LEN | block | cold ms | warm ms | warm GCUPS | MI210 best ms (published) | speed-up vs published (warm) |")
    print("|---|---|---|---|---|---|---|")
    with tempfile.TemporaryDirectory() as d:
        for L in (64, 128, 256, 512, 1024):
            inp = os.path.join(d, "input_%d.txt" % L)
            synth.write_sw_file(inp, synth.sw_pairs(args.alignments, L, L, seed=L))
            for block in (32, 64, 128, 256, 512, 1024):
                res = {}
                for warm in (False, True):
                    ts = [run(exe, inp, os.path.join(d, "o.txt"), block, warm) for _ in range(args.runs)]
                    m = statistics.mean(ts)
                    ci = 1.833 * statistics.stdev(ts) / len(ts) ** 0.5 if len(ts) > 1 else 0.0  # t(0.95, 9)
                    res[warm] = (m, ci)
                g = args.alignments * L * L / (res[True][0] * 1e-3) / 1e9
                print("| %d | %d | %.3f +- %.3f | %.3f +- %.3f | %.0f | %.2f | %.0fx |" % (
                    L, block, res[False][0], res[False][1], res[True][0], res[True][1], g, MI210_BEST_MS[L],
                    MI210_BEST_MS[L] / res[True][0]), flush=True)


if __name__ == "__main__":
    main()
