#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its config, one JSON line on rank 0.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Headline (`value`): Smith-Waterman GCUPS on BASELINE config 2 -- per GPU one batch of 65 536
pairs, 150x150, iid ACGT + 25 % related pairs, int32 affine-gap scores (computed in packed int16
lanes, bit-identical) -- with the packed batch
already resident in HBM when the clock starts.  A step is one pass of the fill over that batch
(agx_sw_batch_launch).  GCUPS counts len_a*len_b with the newline sentinel excluded:
22 500 cells per pair although the kernel fills 151x151.

Second leg, same JSON line under "pairhmm": BASELINE config 3 -- 65 536 (read, haplotype) pairs,
R=100, H=300, fp32 forward (AGX_PHMM_F32_FMA: packed FMA, two haplotypes per lane group) with
double rescue of underflowing pairs -- in pairs/s.

"config4" / "config5": the two 8-GPU configs of BASELINE.json at one GPU's 1/8 shard (131 072 mixed SW
pairs; 32 768 PairHMM pairs R=250 H=500 in bit-identical fp64), a tenth of the steps each.

N > 1: every rank owns its own batch of the same shape (independent pairs shard with no
collective), so scaling is "weak"; torch.distributed (RCCL) is used only for the
barriers and the max-over-ranks of the timed region.

"roofline": HBM bound as BASELINE.json asks; achieved = algorithmic bytes of one launch
(304 B/pair SW, 808 B/pair PairHMM) / mean launch duration from HIP events on
the launch stream.  These kernels are VALU-bound by construction, so the fraction is tiny; the
"valu" sub-object prices the same launch against the integer/fp32 vector issue rate.
"cpu_baseline": the reference C program itself (oracle/_ref, compiled in the authoring
container from the unmodified sources) when present, else the oracle's C port, one core.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SW_PAIRS, SW_LEN = 65536, 150
PH_REGIONS, PH_READS, PH_HAPS, PH_R, PH_H = 64, 64, 16, 100, 300
C4_PAIRS = 1 << 20                                                  # config 4, all 8 GPUs together
C5_REGIONS, C5_READS, C5_HAPS, C5_R, C5_H = 512, 32, 16, 250, 500   # config 5 (262 144 pairs), all 8 GPUs together
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
# Measured wave64 issue rates on this chip (tools/valu_microbench.hip, profiles/r01_valu_microbench.log):
# add/xor/mul class ~65 T lane-op/s; max/max3/cndmask/compare/DPP/fma/f64 class ~38 T lane-op/s.
# packed (v_pk_*) instructions ~37.5 T lane-instr/s = 75 T element-op/s.
VALU_FAST, VALU_SLOW, VALU_PACKED = 65e12, 38e12, 37.5e12


def cpu_baseline_sw(n_pairs):
    """Time the reference SW program (or the oracle port) on the first n_pairs of the rank-0 workload."""
    import accelerating_genomics_amd.synth as synth

    b = synth.sw_pairs(n_pairs, SW_LEN, SW_LEN, seed=2, related_frac=0.25)
    cells = n_pairs * SW_LEN * SW_LEN
    ref = os.path.join(ROOT, "oracle", "_ref", "sw_ref")
    if os.access(ref, os.X_OK):
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "sw.in")
            synth.write_sw_file(path, b)
            t0 = time.perf_counter()
            out = subprocess.run([ref, path], capture_output=True, check=True)
            dt = time.perf_counter() - t0
        assert out.stdout.count(b"Score:") == n_pairs
        kind = "reference"
    else:
        from tests import oracle_api

        orc = oracle_api.load()
        t0 = time.perf_counter()
        orc.sw_batch(b, 0)
        dt = time.perf_counter() - t0
        kind = "port"
    return {"value": cells / dt / 1e9, "unit": "GCUPS", "cores": 1, "kind": kind,
            "sample": "%d of the 65536 config-2 pairs, antidiagonalSmithWaterman.c incl. its text parsing, %.1f s" % (n_pairs, dt)}


def cpu_baseline_sw_multicore(n_procs, pairs_each):
    """BASELINE.md section 3 also asks for an embarrassingly-parallel run: n_procs copies of the reference
    program, one shard each, all started together (the box's CPU share for one GPU is 16 cores)."""
    import accelerating_genomics_amd.synth as synth

    ref = os.path.join(ROOT, "oracle", "_ref", "sw_ref")
    if not os.access(ref, os.X_OK):
        return None
    with tempfile.TemporaryDirectory() as d:
        paths = []
        for k in range(n_procs):
            path = os.path.join(d, "sw%d.in" % k)
            synth.write_sw_file(path, synth.sw_pairs(pairs_each, SW_LEN, SW_LEN, seed=100 + k, related_frac=0.25))
            paths.append(path)
        t0 = time.perf_counter()
        procs = [subprocess.Popen([ref, p], stdout=subprocess.DEVNULL) for p in paths]
        for pr in procs:
            pr.wait()
        dt = time.perf_counter() - t0
    return {"value": n_procs * pairs_each * SW_LEN * SW_LEN / dt / 1e9, "unit": "GCUPS", "cores": n_procs, "kind": "reference",
            "sample": "%d concurrent copies of antidiagonalSmithWaterman.c, %d pairs each, %.1f s" % (n_procs, pairs_each, dt)}


def cpu_baseline_phmm(n_regions):
    import accelerating_genomics_amd.synth as synth

    p = synth.phmm_regions(n_regions, PH_READS, PH_HAPS, PH_R, PH_H, seed=3)
    ref = os.path.join(ROOT, "oracle", "_ref", "phmm_matrix_ref")
    if os.access(ref, os.X_OK):
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "p.in")
            synth.write_phmm_file(path, p)
            t0 = time.perf_counter()
            subprocess.run([ref, path, os.path.join(d, "p.out")], capture_output=True, check=True)
            dt = time.perf_counter() - t0
        kind, what = "reference", "pairHMMmatrix.c (fp64; antidiagsPairHMM.c leaks 24 B/cell, SURVEY.md Q9)"
    else:
        from tests import oracle_api

        orc = oracle_api.load()
        t0 = time.perf_counter()
        orc.phmm_batch(p, 1)
        dt = time.perf_counter() - t0
        kind, what = "port", "oracle antidiag port (fp64)"
    return {"value": p.n_pairs / dt, "unit": "pairs/s", "cores": 1, "kind": kind,
            "sample": "%d of the 65536 config-3 pairs, %s, %.1f s" % (p.n_pairs, what, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-pairs", type=int, default=32768)
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the config 4 / config 5 legs (PMC profiling passes)")
    # rehearsal of the multi-rank control flow on a box with fewer GPUs than ranks (never for numbers):
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--share-device", action="store_true", help="map every rank to device 0")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import accelerating_genomics_amd.api as agx
    import accelerating_genomics_amd.dist as agd
    import accelerating_genomics_amd.synth as synth

    rank, local_rank, world = agd.env_rank()
    multi = world > 1
    if agx.device_count() < 1:
        raise SystemExit("bench.py: no HIP device visible; libagx has no CPU fallback")
    if args.share_device:
        local_rank = 0
    # a launcher that masks devices per rank (HIP_VISIBLE_DEVICES) leaves every rank one device, number 0
    local_rank %= agx.device_count()
    torch.cuda.set_device(local_rank)
    red_dev = "cuda" if args.dist_backend == "nccl" else "cpu"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    n_gpus = world if multi else 1
    if args.gpus != n_gpus and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE %d; reporting n_gpus=%d" % (args.gpus, world, n_gpus), file=sys.stderr)

    ctx = agx.Context(local_rank)

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    def timed(dev):
        for _ in range(args.warmup):
            dev.launch()
        barrier()
        t0 = time.perf_counter()
        ctx.timer_start()
        for _ in range(args.steps):
            dev.launch()
        ev_ms = ctx.timer_stop()  # HIP events on the launch stream (also drains it)
        barrier()
        dt = time.perf_counter() - t0
        return agd.max_over_ranks(dt, device=red_dev), ev_ms / args.steps

    # ---------------- Smith-Waterman, BASELINE config 2 (per rank: its own seed => its own shard)
    sw = synth.sw_pairs(SW_PAIRS, SW_LEN, SW_LEN, seed=2 + 1000 * rank, related_frac=0.25)
    sw_dev = ctx.sw_batch(sw)
    sw_info = sw_dev.info()
    sw_dt, sw_launch_ms = timed(sw_dev)
    sw_scores = sw_dev.scores()
    sw_cells = sw.cells(sentinel=False)  # 65536 * 22500
    sw_gcups = n_gpus * sw_cells * args.steps / sw_dt / 1e9
    sw_bytes = sw.algorithmic_bytes()  # 304 B/pair
    sw_dev.close()

    # ---------------- PairHMM, BASELINE config 3
    ph = synth.phmm_regions(PH_REGIONS, PH_READS, PH_HAPS, PH_R, PH_H, seed=3 + 1000 * rank)
    ph_dev = ctx.phmm_batch(ph, agx.PHMM_F32_FMA)
    ph_info = ph_dev.info()
    ph_dt, ph_launch_ms = timed(ph_dev)
    ph_l, _ = ph_dev.results()
    ph_rescued = ph_dev.info().n_rescued
    ph_rate = n_gpus * ph.n_pairs * args.steps / ph_dt
    ph_bytes = ph.algorithmic_bytes()  # 808 B/pair
    ph_dev.close()

    # ---------------- the two 8-GPU configs of BASELINE.json, each rank its 1/8 shard (a tenth of the steps)
    def timed_few(dev):
        keep = args.steps, args.warmup
        args.steps, args.warmup = max(10, keep[0] // 10), max(2, keep[1] // 10)
        try:
            dt, launch_ms = timed(dev)
            return dt, launch_ms, args.steps
        finally:
            args.steps, args.warmup = keep

    extra = not args.no_extra_configs
    if extra:
        c4 = synth.sw_pairs(C4_PAIRS // 8, 32, 512, seed=4 + 1000 * rank)
        c4_dev = ctx.sw_batch(c4)
        c4_info = c4_dev.info()
        c4_dt, c4_ms, c4_steps = timed_few(c4_dev)
        c4_sum = int(c4_dev.scores().astype(np.int64).sum())
        c4_dev.close()
        c5 = synth.phmm_regions(C5_REGIONS // 8, C5_READS, C5_HAPS, C5_R, C5_H, seed=5 + 1000 * rank)
        c5_dev = ctx.phmm_batch(c5, agx.PHMM_F64)
        c5_info = c5_dev.info()
        c5_dt, c5_ms, c5_steps = timed_few(c5_dev)
        c5_l, _ = c5_dev.results()
        c5_dev.close()

    if rank != 0:
        if multi:
            dist.barrier()
            dist.destroy_process_group()
        return

    traffic = None
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tfile):
        traffic = json.load(open(tfile))

    def roof(alg_bytes, launch_ms, key):
        ach = alg_bytes / (launch_ms * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": (traffic or {}).get(key), "launch_ms": launch_ms, "algorithmic_bytes_per_launch": alg_bytes}

    out = {
        "metric": "Smith-Waterman affine-gap score-only GCUPS (config 2: 65536 pairs 150x150 per GPU)",
        "value": sw_gcups, "unit": "GCUPS", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": sw_dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int16x2 lanes, int32 scores", "data": "synthetic",
        "config": {"workload": "BASELINE config 2: 65536 SW pairs 150x150 per GPU, iid ACGT + 25% related, newline sentinel aligned as the reference does",
                   "pairs_per_gpu": SW_PAIRS, "len": SW_LEN, "cells_counted_per_pair": SW_LEN * SW_LEN,
                   "parallelism": "pairs sharded per GPU, no collective"},
        "roofline": roof(sw_bytes, sw_launch_ms, "sw_fill"),
        "sw": {"waves": sw_info.n_waves, "launches_per_step": sw_info.n_launches,
               "useful_cell_fraction": sw_info.cells / max(1, sw_info.padded_cells),
               "valu": {"ops_per_cell": "12 packed int16 instructions per 2 cells (v_pk_add/max/min_u16/sub_u16 clamp + xor)",
                        "frac_of_instruction_mix_ceiling": sw_info.padded_cells * (6.0 / VALU_PACKED) / (sw_launch_ms * 1e-3)},
               "score_checksum": int(sw_scores.astype(np.int64).sum())},
        "pairhmm": {
            "metric": "PairHMM forward pairs/s (config 3: 65536 pairs R=100 H=300 fp32 per GPU)",
            "value": ph_rate, "unit": "pairs/s", "ms_per_step": ph_dt / args.steps * 1e3, "dtype": "f32 (two haplotypes per lane group, packed FMA; double rescue)",
            "gcups": n_gpus * ph.cells() * args.steps / ph_dt / 1e9, "rescued_in_f64": int(ph_rescued),
            "waves": ph_info.n_waves, "launches_per_step": ph_info.n_launches,
            "useful_cell_fraction": ph_info.cells / max(1, ph_info.padded_cells),
            "roofline": roof(ph_bytes, ph_launch_ms, "phmm_fill"),
            "valu": {"ops_per_cell": "8 packed fp32 instructions + 2 compares + 2 selects per 2 cells",
                     "frac_of_instruction_mix_ceiling": ph_info.padded_cells * (4 / VALU_PACKED + 2 / VALU_SLOW) / (ph_launch_ms * 1e-3)},
            "log10_checksum": float(ph_l.sum()),
        },
    }
    if extra:
        out["config4"] = {
            "metric": "Smith-Waterman GCUPS, mixed lengths 32-512 (config 4: 1 048 576 pairs over 8 GPUs; %d pairs per GPU here)" % (C4_PAIRS // 8),
            "value": n_gpus * c4.cells(sentinel=False) * c4_steps / c4_dt / 1e9, "unit": "GCUPS", "steps": c4_steps,
            "ms_per_step": c4_dt / c4_steps * 1e3, "launch_ms": c4_ms, "launches_per_step": c4_info.n_launches,
            "useful_cell_fraction": c4_info.cells / max(1, c4_info.padded_cells), "score_checksum": c4_sum}
        out["config5"] = {
            "metric": "PairHMM forward pairs/s, fp64 in the reference's operation order (config 5: 262 144 pairs R=250 H=500 over 8 GPUs; %d pairs per GPU here)" % c5.n_pairs,
            "value": n_gpus * c5.n_pairs * c5_steps / c5_dt, "unit": "pairs/s", "dtype": "f64", "steps": c5_steps,
            "ms_per_step": c5_dt / c5_steps * 1e3, "launch_ms": c5_ms, "gcups": n_gpus * c5.cells() * c5_steps / c5_dt / 1e9,
            "launches_per_step": c5_info.n_launches, "useful_cell_fraction": c5_info.cells / max(1, c5_info.padded_cells),
            "log10_checksum": float(c5_l.sum())}
    if not args.no_cpu_baseline and n_gpus == 1:  # the CPU baseline is reported at N=1 only
        out["cpu_baseline"] = cpu_baseline_sw(args.cpu_sample_pairs)
        out["pairhmm"]["cpu_baseline"] = cpu_baseline_phmm(max(1, args.cpu_sample_pairs // (PH_READS * PH_HAPS)))
        out["cpu_baseline"]["host_cores_available"] = os.cpu_count()
        multi_cpu = cpu_baseline_sw_multicore(min(16, os.cpu_count() or 1), 8192)
        if multi_cpu:
            out["cpu_baseline_multicore"] = multi_cpu
    print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
