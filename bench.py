#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its config, one JSON line on rank 0.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A STEP is one pass of the hot path over one batch in the reference's own timing window
(smithWaterman/hipvers.cpp:475-483): kernel launch -> results resident in HOST memory, with
the inputs already resident in HBM when the clock starts.  Every step is clocked on its own (median and
min are reported beside the mean); the launch inside it is bracketed by HIP events on the launch stream
(kernel-only duration, the roofline's denominator).  Before the W untimed warm-up steps the device is
warmed BY TIME (--warm-seconds of back-to-back launches), so a short run (--steps 20) and a long one
(--steps 500) report the same figures.

Headline (`value`): Smith-Waterman GCUPS on BASELINE config 2 -- per GPU one batch of 65 536 pairs, 150x150,
iid ACGT + 25 % related pairs, int32 affine-gap scores computed in packed int16 lanes (bit-identical; the
"sw_int32" leg times the int32 kernel on the same batch).  GCUPS counts len_a*len_b with the newline
sentinel excluded: 22 500 cells per pair although the kernel fills 151x151.

"pairhmm": BASELINE config 3 -- 65 536 (read, haplotype) pairs, R=100, H=300, fp32 forward (packed FMA, two
haplotypes per lane group) with double rescue of underflowing pairs -- in pairs/s, same window.

"config4" / "config5": the two 8-GPU configs of BASELINE.json.  Weak leg: every rank its own 1/8-size shard
(131 072 mixed SW pairs; 32 768 PairHMM pairs R=250 H=500 in bit-identical fp64).  "total" (strong scaling):
ONE host batch of the full size (1 048 576 pairs; 262 144 pairs), identical on every rank, cut into
world-size contiguous shards by cells exactly as agx_*_devices cut (agx_sw_shard_cuts / agx_phmm_shard_cuts);
every rank fills its shard, the clock is the max over ranks; "per_rank" lists each shard's cells and launch
time.  "multi_one_process": the same batch through agx_*_multi(N) from rank 0 alone, host buffers in, results
out (needs all N devices visible to rank 0; skipped otherwise).

"one_shot": host-inclusive calls (agx_sw_score / agx_phmm_forward: plan + H2D + fill + D2H) on config 2 / 3,
pageable and page-locked source buffers.  Never the headline.

"roofline": HBM bound as BASELINE.json asks; achieved = algorithmic bytes of one launch (304 B/pair SW,
808 / 1758 B/pair PairHMM) / the launch's mean duration from the HIP events of the timed
steps.  These kernels are VALU-bound by construction, so the fraction is tiny; "valu" prices the same
launch against the measured vector issue rate.  "traffic" is NOT measured in this run: it is the PMC figure
of the rocprofv3 pass named in "traffic_source".
"cpu_baseline": the reference C program itself (oracle/_ref, compiled in the authoring container from the
unmodified sources) when present, else the oracle's C port; one core, a bounded sample of the same workload.
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
# Measured wave64 issue cost on this chip (tools/valu_microbench2.hip, profiles/r02_valu_microbench2.log): every
# v_pk_* instruction 4.2 cycles per SIMD; v_add/sub/xor_u32 2.4 in a pure stream but 4.2 beside packed ones
# (profiles/r02_valu_microbench3.log), so every instruction of the cell is priced at the packed rate.
VALU_PACKED = 37.5e12  # lane-instructions / s
WINDOW = "kernel launch -> results resident in (page-locked) host memory, inputs resident in HBM (hipvers.cpp:475-483)"


# ----------------------------------------------------------------------------------------- CPU baselines

def _sw_ref_or_port(b, label):
    """Time the reference SW program (or the oracle port) on batch b, one core."""
    import accelerating_genomics_amd.synth as synth

    cells = b.cells(sentinel=False)
    ref = os.path.join(ROOT, "oracle", "_ref", "sw_ref")
    if os.access(ref, os.X_OK):
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "sw.in")
            synth.write_sw_file(path, b)
            t0 = time.perf_counter()
            out = subprocess.run([ref, path], capture_output=True, check=True)
            dt = time.perf_counter() - t0
        assert out.stdout.count(b"Score:") == b.n_pairs
        kind, what = "reference", "antidiagonalSmithWaterman.c incl. its text parsing"
    else:
        from tests import oracle_api

        orc = oracle_api.load()
        t0 = time.perf_counter()
        orc.sw_batch(b, 0)
        dt = time.perf_counter() - t0
        kind, what = "port", "oracle anti-diagonal port"
    return {"value": cells / dt / 1e9, "unit": "GCUPS", "cores": 1, "kind": kind, "sample": "%s, %s, %.1f s" % (label, what, dt)}


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


def _phmm_ref_or_port(p, label, program="phmm_matrix_ref"):
    """Time a reference PairHMM program (or the oracle port) on the regions p, one core.  `program`:
    phmm_matrix_ref = pairHMMmatrix.c, phmm_antidiag_ref = antidiagsPairHMM.c (the program north_star names; it
    leaks 24 B per cell, antidiagsPairHMM.c:144-151, so its sample is kept to a few hundred pairs)."""
    import accelerating_genomics_amd.synth as synth

    ref = os.path.join(ROOT, "oracle", "_ref", program)
    if os.access(ref, os.X_OK):
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "p.in")
            synth.write_phmm_file(path, p)
            t0 = time.perf_counter()
            subprocess.run([ref, path, os.path.join(d, "p.out")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            dt = time.perf_counter() - t0
        kind = "reference"
        what = ("antidiagsPairHMM.c itself (fp64; leaks 24 B per cell, so the sample is small)" if program == "phmm_antidiag_ref"
                else "pairHMMmatrix.c (fp64; same numbers as antidiagsPairHMM.c, no leak)")
    else:
        from tests import oracle_api

        orc = oracle_api.load()
        t0 = time.perf_counter()
        orc.phmm_batch(p, 1)
        dt = time.perf_counter() - t0
        kind, what = "port", "oracle antidiag port (fp64)"
    return {"value": p.n_pairs / dt, "unit": "pairs/s", "cores": 1, "kind": kind, "gcups": p.cells() / dt / 1e9,
            "sample": "%s, %s, %.1f s" % (label, what, dt)}


def cpu_baseline_phmm_multicore(n_procs, regions_each):
    """n_procs concurrent copies of pairHMMmatrix.c, each on its own file of config-3 regions."""
    import accelerating_genomics_amd.synth as synth

    ref = os.path.join(ROOT, "oracle", "_ref", "phmm_matrix_ref")
    if not os.access(ref, os.X_OK):
        return None
    with tempfile.TemporaryDirectory() as d:
        paths, pairs = [], 0
        for k in range(n_procs):
            p = synth.phmm_regions(regions_each, PH_READS, PH_HAPS, PH_R, PH_H, seed=300 + k)
            pairs += p.n_pairs
            path = os.path.join(d, "p%d.in" % k)
            synth.write_phmm_file(path, p)
            paths.append(path)
        t0 = time.perf_counter()
        procs = [subprocess.Popen([ref, q, q + ".out"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for q in paths]
        for pr in procs:
            pr.wait()
        dt = time.perf_counter() - t0
    return {"value": pairs / dt, "unit": "pairs/s", "cores": n_procs, "kind": "reference",
            "sample": "%d concurrent copies of pairHMMmatrix.c, %d config-3 pairs each, %.1f s" % (n_procs, pairs // n_procs, dt)}


def host_cores():
    """Cores this process may use (the box's share): its affinity mask, cut by the cgroup's CPU quota where one is set."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]           # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())      # cgroup v1
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = min(n, max(1, int(quota / period + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--warm-seconds", type=float, default=0.5, help="back-to-back launches before every timed leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true", help="headline legs only (PMC profiling passes)")
    # The control plane (barriers, max of the ranks' clocks, the per-rank table: host scalars only) runs on gloo by
    # default; "nccl" (= RCCL) puts the same three calls on device tensors.  The data path has no collective.
    ap.add_argument("--dist-backend", default="gloo", choices=["nccl", "gloo"])
    # rehearsal of the multi-rank control flow on a box with fewer GPUs than ranks (never for numbers):
    ap.add_argument("--share-device", action="store_true", help="map every rank to device 0")
    return ap.parse_args(argv)


def self_launch(args):
    """`python bench.py --gpus N` run bare (no launcher, WORLD_SIZE unset): start the N ranks here -- one process per
    GPU, the environment torch.distributed.run would give them -- relay rank 0's one JSON line and leave with the worst
    return code.  This runs BEFORE torch or libagx are imported: the parent never makes a HIP call (a process that has
    touched the GPU must not fork/exec launchers), it only waits."""
    import socket

    assert "torch" not in sys.modules and "accelerating_genomics_amd.api" not in sys.modules
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    try:
        maps = open("/proc/self/maps").read()
    except OSError:
        maps = ""
    hip_mapped = any(lib in maps for lib in ("libamdhip64", "libagx", "libhsa-runtime64"))
    assert not hip_mapped, "the launcher process has a HIP library mapped"
    print("bench.py: WORLD_SIZE unset and --gpus %d: starting %d ranks (parent has imported torch: %s, libagx: %s, HIP libraries mapped: %s), "
          "rendezvous 127.0.0.1:%d" % (args.gpus, args.gpus, "torch" in sys.modules, "accelerating_genomics_amd.api" in sys.modules, hip_mapped, port),
          file=sys.stderr, flush=True)
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), AGX_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # rank 0's stdout is the JSON line; the other ranks print nothing there, whatever they do print goes to stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno(), cwd=os.getcwd()))
    # rank 0's pipe is drained on a thread while the parent watches all ranks: when one of them fails the others would
    # wait in a barrier for the group's timeout, so they get 30 s and are then ended (by their own PIDs)
    import threading

    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed_at = None
    while any(p.poll() is None for p in procs):
        if failed_at is None and any(p.poll() not in (None, 0) for p in procs):
            failed_at = time.monotonic()
        if failed_at is not None and time.monotonic() - failed_at > 30.0:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        time.sleep(0.05)
    codes = [p.wait() for p in procs]
    reader.join(timeout=10.0)
    sys.stdout.write(b"".join(out0).decode(errors="replace"))
    sys.stdout.flush()
    worst = max((abs(c) for c in codes), default=0)
    if worst:
        print("bench.py: rank return codes %s" % codes, file=sys.stderr)
    raise SystemExit(min(worst, 255))


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    if args.gpus > 1 and int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit("bench.py: --gpus %d under a launcher with WORLD_SIZE=%s: start as many ranks as GPUs asked for"
                         % (args.gpus, os.environ["WORLD_SIZE"]))

    import torch
    import torch.distributed as dist

    import accelerating_genomics_amd.api as agx
    import accelerating_genomics_amd.dist as agd
    import accelerating_genomics_amd.synth as synth

    rank, local_rank, world = agd.env_rank()
    # a launcher (torch.distributed.run, or self_launch above) set WORLD_SIZE: the process group comes up even for one rank
    multi = "WORLD_SIZE" in os.environ and ("MASTER_PORT" in os.environ)
    n_dev = agx.device_count()
    if n_dev < 1:
        raise SystemExit("bench.py: rank %d of %d: no HIP device visible; libagx has no CPU fallback" % (rank, world))
    masked = any(os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))
    if args.share_device:
        local_rank = 0
    elif local_rank >= n_dev:
        if masked and n_dev == 1:
            local_rank = 0  # a launcher that masks devices per rank leaves every rank one device, number 0
        else:
            raise SystemExit("bench.py: rank %d wants device %d but only %d visible; a number measured with ranks sharing a GPU is not a "
                             "%d-GPU number (--share-device rehearses the control flow)" % (rank, local_rank, n_dev, world))
    torch.cuda.set_device(local_rank)
    red_dev = "cuda" if args.dist_backend == "nccl" else "cpu"
    cpu_group = None
    control = {"backend": "none", "world_size": 1}
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # Gloo announces its connections on stdout ("[Gloo] Rank 0 is connected to ..."): stdout carries the ONE
        # JSON line and nothing else, so file descriptor 1 points at stderr while the groups come up.
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if args.dist_backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
                cpu_group = dist.new_group(backend="gloo")  # waits that must not park a kernel on the GPUs
                dist.barrier(group=cpu_group)
            else:
                dist.init_process_group("gloo")
                dist.barrier()
            # the three calls the run depends on, once, with values that can be checked
            probe = torch.tensor([float(rank + 1)], dtype=torch.float64, device=red_dev)
            dist.all_reduce(probe, op=dist.ReduceOp.MAX)
            rows = [torch.zeros(2, dtype=torch.float64, device=red_dev) for _ in range(world)]
            dist.all_gather(rows, torch.tensor([float(rank), float(local_rank)], dtype=torch.float64, device=red_dev))
            assert float(probe.item()) == float(world) and [int(r[0].item()) for r in rows] == list(range(world)), "control plane self-check"
            # two ranks on one GPU would report a number that is not an N-GPU number: every rank names the device it took
            # (the masks that hide devices from it, the ordinal inside them, the device's UUID where torch tells it)
            try:
                uuid = str(torch.cuda.get_device_properties(local_rank).uuid)
            except Exception:
                uuid = ""
            me = (os.uname().nodename, tuple(os.environ.get(k, "") for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")), local_rank, uuid)
            names = [None] * world
            dist.all_gather_object(names, me, group=cpu_group)
            if len(set(names)) < world and not args.share_device:
                raise SystemExit("bench.py: rank %d: ranks share a GPU (%s); a number measured that way is not a %d-GPU number "
                                 "(--share-device rehearses the control flow)" % (rank, names, world))
            control = {"backend": args.dist_backend, "world_size": world, "tensors_on": red_dev,
                       "self_check": "all_reduce(MAX) and all_gather returned the expected values",
                       "devices_of_ranks": [int(r[1].item()) for r in rows], "self_launched": bool(os.environ.get("AGX_BENCH_SELF_LAUNCHED"))}
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    n_gpus = world if multi else 1
    if args.gpus != n_gpus:
        raise SystemExit("bench.py: --gpus %d but %d rank(s) are running" % (args.gpus, n_gpus))

    ctx = agx.Context(local_rank)

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    def warm(dev):
        """Clocks and queues settle over a few hundred ms of work: warm by time, not by count."""
        t_end = time.perf_counter() + args.warm_seconds
        while time.perf_counter() < t_end:
            for _ in range(8):
                dev.launch()
            ctx.sync()

    def timed(dev, fetch, steps, warmup, unbind=None):
        """-> dict: wall clock over `steps` steps (max over ranks), per-step host times, per-step kernel times, and
        the kernel-only figure from back-to-back launches."""
        warm(dev)
        for _ in range(warmup):
            dev.launch()
            fetch()
        per = np.empty(steps)
        kern = []
        barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            ta = time.perf_counter()
            if k % 4 == 0:  # HIP events around the launch on its stream, on every fourth step of the timed region: the
                ctx.timer_start()  # two records and the query cost 6-8 us of host time, 3-4 % of a 0.2 ms step
                dev.launch()
                ctx.timer_mark()
                fetch()
                per[k] = time.perf_counter() - ta
                kern.append(ctx.timer_elapsed())
            else:
                dev.launch()
                fetch()  # waits for the stream, copies the results into host memory
                per[k] = time.perf_counter() - ta
        kern = np.array(kern)
        # every step ended with its results on the host (fetch() waits for the stream), so this rank's K steps are
        # complete here: its clock stops, THEN the ranks meet and the slowest one's time is the job's
        torch.cuda.synchronize()
        ctx.sync()
        mine = time.perf_counter() - t0
        barrier()
        dt = agd.max_over_ranks(mine, device=red_dev)
        if unbind is not None:  # the kernel-only figure is the fill with its results left in HBM
            unbind()
        ctx.timer_start()
        for _ in range(steps):
            dev.launch()
        b2b = ctx.timer_stop() / steps
        return {"dt": dt, "steps": steps, "step_ms": {"median": float(np.median(per)) * 1e3, "min": float(per.min()) * 1e3,
                                                     "mean": float(per.mean()) * 1e3, "max": float(per.max()) * 1e3},
                "launch_ms": float(kern.mean()), "launch_ms_min": float(kern.min()), "back_to_back_launch_ms": b2b}

    few = lambda: (max(20, args.steps // 5), max(3, args.warmup // 5))  # (the other legs: 20 steps -- one slow step among ten moved a leg by 5-8 % from run to run)
    traffic = {}
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tfile):
        traffic = json.load(open(tfile))

    def roof(alg_bytes, launch_ms, key, in_hbm_ms=None):
        ach = alg_bytes / (launch_ms * 1e-3) / 1e9
        r = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
             "traffic": traffic.get(key), "traffic_source": traffic.get("_source") if traffic.get(key) is not None else None,
             "launch_ms": launch_ms, "algorithmic_bytes_per_launch": alg_bytes}
        if in_hbm_ms is not None:
            # the timed launches deliver their results over PCIe into the caller's page-locked array themselves (bound results:
            # no copy / log10 kernel behind the fill, the step is shorter, the launch a few microseconds longer); the same
            # fill with its results left in HBM, back to back:
            r["launch_ms_results_left_in_hbm"] = in_hbm_ms
            r["frac_results_left_in_hbm"] = alg_bytes / (in_hbm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        return r

    def leg(t, units, per_unit_name):
        """Throughput fields of one leg: whole-job units over the max-over-ranks wall clock of its steps."""
        return {"value": n_gpus * units * t["steps"] / t["dt"], "steps": t["steps"], "ms_per_step": t["dt"] / t["steps"] * 1e3,
                "step_ms": t["step_ms"], "window": WINDOW,
                "kernel_only": {per_unit_name: n_gpus * units / (t["back_to_back_launch_ms"] * 1e-3),
                                "launch_ms": t["back_to_back_launch_ms"], "what": "back-to-back launches, HIP events, results stay in HBM (unbound)"}}

    # ---------------- Smith-Waterman, BASELINE config 2 (per rank: its own seed => its own shard)
    sw = synth.sw_pairs(SW_PAIRS, SW_LEN, SW_LEN, seed=2 + 1000 * rank, related_frac=0.25)
    sw_out = agx.host_array(sw.n_pairs, np.int32)  # results land in page-locked host memory: one DMA, no staging copy
    sw_dev = ctx.sw_batch(sw)
    sw_dev.bind_scores(sw_out)  # a batch in file order writes its scores into the page-locked array itself: no copy kernel behind the fill
    sw_info = sw_dev.info()
    sw_t = timed(sw_dev, lambda: sw_dev.scores(sw_out), args.steps, args.warmup, unbind=lambda: sw_dev.bind_scores(None))
    sw_sum = int(sw_out.astype(np.int64).sum())
    sw_dev.close()
    sw_cells = sw.cells(sentinel=False)  # 65536 * 22500
    # the same batch through the int32 kernel (BASELINE config 2 as worded); scores must be the same
    ctx.set_option(agx.OPT_SW_KERNEL, agx.SW_KERNEL_INT32)
    i32_dev = ctx.sw_batch(sw)
    ctx.set_option(agx.OPT_SW_KERNEL, agx.SW_KERNEL_AUTO)
    i32_info = i32_dev.info()
    i32_out = agx.host_array(sw.n_pairs, np.int32)
    i32_dev.bind_scores(i32_out)
    i32_t = timed(i32_dev, lambda: i32_dev.scores(i32_out), *few(), unbind=lambda: i32_dev.bind_scores(None))
    i32_same = bool(np.array_equal(i32_out, sw_out))
    i32_dev.close()

    # ---------------- PairHMM, BASELINE config 3
    ph = synth.phmm_regions(PH_REGIONS, PH_READS, PH_HAPS, PH_R, PH_H, seed=3 + 1000 * rank)
    ph_out = (agx.host_array(ph.n_pairs, np.float64), None)
    ph_dev = ctx.phmm_batch(ph, agx.PHMM_F32_FMA)
    ph_dev.bind_results(ph_out[0])  # a batch in output order writes its log10 likelihoods into the page-locked array itself
    ph_info = ph_dev.info()
    ph_t = timed(ph_dev, lambda: ph_dev.results(ph_out, want_sums=False), args.steps, args.warmup, unbind=lambda: ph_dev.bind_results(None))
    ph_rescued = ph_dev.info().n_rescued
    ph_sum = float(ph_out[0].sum())
    ph_dev.close()

    extra = not args.no_extra_configs
    if extra:
        # ---------------- weak legs of the two 8-GPU configs: every rank its own 1/8-size shard
        c4 = synth.sw_pairs(C4_PAIRS // 8, 32, 512, seed=4 + 1000 * rank)
        c4_out = agx.host_array(c4.n_pairs, np.int32)
        c4_dev = ctx.sw_batch(c4)
        c4_info = c4_dev.info()
        c4_t = timed(c4_dev, lambda: c4_dev.scores(c4_out), *few())
        c4_sum = int(c4_out.astype(np.int64).sum())
        c4_dev.close()
        c5 = synth.phmm_regions(C5_REGIONS // 8, C5_READS, C5_HAPS, C5_R, C5_H, seed=5 + 1000 * rank)
        c5_out = (agx.host_array(c5.n_pairs, np.float64), None)
        c5_dev = ctx.phmm_batch(c5, agx.PHMM_F64)
        c5_info = c5_dev.info()
        c5_t = timed(c5_dev, lambda: c5_dev.results(c5_out, want_sums=False), *few())
        c5_sum = float(c5_out[0].sum())
        c5_dev.close()
        # the same shard with explicit fma (AGX_PHMM_F64_FMA): inside the tolerance config 5 states (1e-12), not bit-identical
        c5m_out = (agx.host_array(c5.n_pairs, np.float64), None)
        c5m_dev = ctx.phmm_batch(c5, agx.PHMM_F64_FMA)
        c5m_t = timed(c5m_dev, lambda: c5m_dev.results(c5m_out, want_sums=False), *few())
        c5m_dev.close()
        c5m_rel = float(np.max(np.abs(c5m_out[0] - c5_out[0]) / np.maximum(np.abs(c5_out[0]), 1e-300)))

        # ---------------- the reference's own corpus shape: pairHMM/test_set/10s.in (7 regions, 3550 pairs, reads of 10-247
        # bases against haplotypes of 41-263; committed as tests/golden/phmm_10s.in) with its regions repeated 19 times
        # -- 67 450 pairs, 133 regions -- in bit-identical double and in packed float; and the mixed SW golden file's
        # 256 pairs (lengths 32-512) 256 times over.  Mixed regions are where the PairHMM planner pads most.
        corpus_legs = {}
        gold = os.path.join(ROOT, "tests", "golden")
        if os.path.exists(os.path.join(gold, "phmm_10s.in")):
            corpus = synth.phmm_repeat(synth.parse_phmm_text(open(os.path.join(gold, "phmm_10s.in"), "rb").read()), 19)
            for prec, name in ((agx.PHMM_F64, "f64"), (agx.PHMM_F32_FMA, "f32_fma")):
                co_out = (agx.host_array(corpus.n_pairs, np.float64), None)
                co_dev = ctx.phmm_batch(corpus, prec)
                co_info = co_dev.info()
                co_t = timed(co_dev, lambda: co_dev.results(co_out, want_sums=False), *few())
                co_info = co_dev.info()
                co_dev.close()
                corpus_legs[name] = {"pairs_per_s": n_gpus * corpus.n_pairs * co_t["steps"] / co_t["dt"], "ms_per_step": co_t["dt"] / co_t["steps"] * 1e3,
                                     "cells_per_s": n_gpus * corpus.cells() * co_t["steps"] / co_t["dt"],
                                     "kernel_only_ms": co_t["back_to_back_launch_ms"], "launches_per_step": co_info.n_launches, "waves": co_info.n_waves,
                                     "useful_cell_fraction": co_info.cells / max(1, co_info.padded_cells), "rescued_in_f64": int(co_info.n_rescued),
                                     "log10_checksum": float(co_out[0].sum())}
            corpus_legs["pairs"], corpus_legs["regions"], corpus_legs["cells"] = corpus.n_pairs, corpus.n_regions, corpus.cells()
            corpus_legs["what"] = "tests/golden/phmm_10s.in (= the reference's pairHMM/test_set/10s.in) with its 7 regions repeated 19 times, per GPU"
        if os.path.exists(os.path.join(gold, "sw_mixed.in")):
            _, swm, _ = agx.read_sw_text(os.path.join(gold, "sw_mixed.in"))
            swm = swm.subset(np.tile(np.arange(swm.n_pairs), 256))
            sm_out = agx.host_array(swm.n_pairs, np.int32)
            sm_dev = ctx.sw_batch(swm)
            sm_info = sm_dev.info()
            sm_t = timed(sm_dev, lambda: sm_dev.scores(sm_out), *few())
            sm_dev.close()
            corpus_legs["sw_mixed"] = {"gcups": n_gpus * swm.cells(sentinel=False) / 1e9 * sm_t["steps"] / sm_t["dt"], "ms_per_step": sm_t["dt"] / sm_t["steps"] * 1e3,
                                       "pairs": swm.n_pairs, "kernel_only_ms": sm_t["back_to_back_launch_ms"], "launches_per_step": sm_info.n_launches,
                                       "useful_cell_fraction": sm_info.cells / max(1, sm_info.padded_cells), "planned_on_device": int(sm_info.planned_on_device),
                                       "what": "tests/golden/sw_mixed.in (256 pairs, lengths 32-512, newline sentinels as the reference keeps them) 256 times over, per GPU"}

        # ---------------- strong legs: ONE host batch of the full size, the same on every rank, cut by cells
        def per_rank_table(pairs, cells, t):
            """Every rank's shard and launch time, gathered to all ranks (6 numbers each)."""
            mine = torch.tensor([rank, pairs, cells, t["launch_ms"], t["launch_ms_min"], t["step_ms"]["median"]], dtype=torch.float64,
                                device=red_dev)
            rows = [torch.zeros_like(mine) for _ in range(world)] if multi else [mine]
            if multi:
                dist.all_gather(rows, mine)
            return [{"rank": int(r[0]), "pairs": int(r[1]), "cells": int(r[2]), "launch_ms": float(r[3]), "launch_ms_min": float(r[4]),
                     "step_ms_median": float(r[5])} for r in (x.cpu() for x in rows)]

        c4f = synth.sw_pairs(C4_PAIRS, 32, 512, seed=4)
        cut4 = agx.sw_shard_cuts(c4f, world)
        mine4 = c4f.subset(np.arange(cut4[rank], cut4[rank + 1]))
        s4_out = agx.host_array(mine4.n_pairs, np.int32)
        s4_dev = ctx.sw_batch(mine4)
        s4_info = s4_dev.info()
        s4_t = timed(s4_dev, lambda: s4_dev.scores(s4_out), *few())
        s4_dev.close()
        s4_rows = per_rank_table(mine4.n_pairs, mine4.cells(sentinel=False), s4_t)
        c5f = synth.phmm_regions(C5_REGIONS, C5_READS, C5_HAPS, C5_R, C5_H, seed=5)
        cut5 = agx.phmm_shard_cuts(c5f, world)
        mine5 = c5f.regions(int(cut5[rank]), int(cut5[rank + 1]))
        s5_out = (agx.host_array(mine5.n_pairs, np.float64), None)
        s5_dev = ctx.phmm_batch(mine5, agx.PHMM_F64)
        s5_t = timed(s5_dev, lambda: s5_dev.results(s5_out, want_sums=False), *few())
        s5_dev.close()
        s5_rows = per_rank_table(mine5.n_pairs, mine5.cells(), s5_t)
        # ... and through ONE process driving all the devices (agx_*_multi): rank 0 alone, the others wait on the CPU
        multi_one = {"skipped": "rank 0 sees %d device(s), %d needed" % (agx.device_count(), world)}
        if multi:
            dist.barrier(group=cpu_group) if cpu_group is not None else dist.barrier()
        if rank == 0 and agx.device_count() >= world and not args.share_device:
            def best_of(fn, reps=3):
                fn()  # contexts of the other devices, their pools
                ts = []
                for _ in range(reps):
                    t0 = time.perf_counter()
                    fn()
                    ts.append(time.perf_counter() - t0)
                return min(ts), float(np.median(ts))
            t_sw, t_sw_med = best_of(lambda: agx.sw_score_multi(c4f, world))
            # the same from page-locked buffers (agx_host_alloc): the upload is then one DMA per piece straight from the
            # caller's memory instead of a staged copy, and the call is bound by PCIe (573 MB) rather than by host memcpy
            c4p = synth.SWBatch(agx.host_array(c4f.bases.size, np.uint8), agx.host_array(c4f.off.size, np.uint64), agx.host_array(c4f.len.size, np.uint32))
            c4p.bases[:], c4p.off[:], c4p.len[:] = c4f.bases, c4f.off, c4f.len
            t_swp, t_swp_med = best_of(lambda: agx.sw_score_multi(c4p, world))
            t_ph, t_ph_med = best_of(lambda: agx.phmm_forward_multi(c5f, agx.PHMM_F64, world))
            multi_one = {"n_devices": world, "what": "agx_sw_score_multi / agx_phmm_forward_multi from one process: host buffers in, results out "
                         "(plan + H2D + fill + D2H per device), best of 3",
                         "config4": {"ms": t_sw * 1e3, "ms_median": t_sw_med * 1e3, "gcups_host_inclusive": c4f.cells(sentinel=False) / t_sw / 1e9},
                         "config4_page_locked_source": {"ms": t_swp * 1e3, "ms_median": t_swp_med * 1e3, "gcups_host_inclusive": c4f.cells(sentinel=False) / t_swp / 1e9},
                         "config5": {"ms": t_ph * 1e3, "ms_median": t_ph_med * 1e3, "pairs_per_s_host_inclusive": c5f.n_pairs / t_ph}}
        if multi:
            dist.barrier(group=cpu_group) if cpu_group is not None else dist.barrier()

    # ---------------- host-inclusive one-shot calls (rank 0 reports its own; never the headline)
    def one_shot(fn, reps):
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return {"ms_median": float(np.median(ts)) * 1e3, "ms_min": min(ts) * 1e3, "reps": reps}

    if extra and rank == 0:
        reps = few()[0]
        pinned = synth.SWBatch(agx.host_array(sw.bases.size, np.uint8), agx.host_array(sw.off.size, np.uint64), agx.host_array(sw.len.size, np.uint32))
        pinned.bases[:], pinned.off[:], pinned.len[:] = sw.bases, sw.off, sw.len
        os_sw, os_sw_pin = one_shot(lambda: ctx.sw_score(sw), reps), one_shot(lambda: ctx.sw_score(pinned), reps)
        os_ph = one_shot(lambda: ctx.phmm_forward(ph, agx.PHMM_F32_FMA), reps)
        one = {"what": "agx_sw_score / agx_phmm_forward: host buffers in, results out (plan + H2D + fill + D2H); PCIe-inclusive, never the headline",
               "config2_pageable": dict(os_sw, gcups=sw_cells / (os_sw["ms_min"] * 1e-3) / 1e9),
               "config2_pinned": dict(os_sw_pin, gcups=sw_cells / (os_sw_pin["ms_min"] * 1e-3) / 1e9),
               "config3_pageable": dict(os_ph, pairs_per_s=ph.n_pairs / (os_ph["ms_min"] * 1e-3))}

    if rank != 0:
        if multi:
            dist.barrier()
            dist.destroy_process_group()
        return

    sw_leg = leg(sw_t, sw_cells / 1e9, "gcups")
    ph_leg = leg(ph_t, ph.n_pairs, "pairs_per_s")
    i32_leg = leg(i32_t, sw_cells / 1e9, "gcups")
    out = {
        "metric": "Smith-Waterman affine-gap score-only GCUPS (config 2: 65536 pairs 150x150 per GPU)",
        "value": sw_leg["value"], "unit": "GCUPS", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": sw_leg["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int16x2 lanes, int32 scores", "data": "synthetic",
        "config": {"workload": "BASELINE config 2: 65536 SW pairs 150x150 per GPU, iid ACGT + 25% related, newline sentinel aligned as the reference does",
                   "pairs_per_gpu": SW_PAIRS, "len": SW_LEN, "cells_counted_per_pair": SW_LEN * SW_LEN,
                   "parallelism": "pairs sharded per GPU, no collective", "window": WINDOW,
                   "warm_seconds_before_each_leg": args.warm_seconds},
        "step_ms": sw_leg["step_ms"], "kernel_only": sw_leg["kernel_only"],
        "roofline": roof(sw.algorithmic_bytes(), sw_t["launch_ms"], "sw_fill", sw_t["back_to_back_launch_ms"]),
        "sw": {"waves": sw_info.n_waves, "launches_per_step": sw_info.n_launches,
               "useful_cell_fraction": sw_info.cells / max(1, sw_info.padded_cells),
               "valu": {"ops_per_cell": "6.75 instructions per 2 cells (DNA-coded rising cell with column classes): v_pk_maximum3_f16 x2.5, v_pk_max_u16, v_perm_b32, v_add3_u32, v_sub_u32 x1.25",
                        "frac_of_instruction_mix_ceiling": sw_info.padded_cells * (6.75 / 2 / VALU_PACKED) / (sw_t["back_to_back_launch_ms"] * 1e-3),
                        "priced_on": "kernel_only.launch_ms (the fill with its results left in HBM)"},
               "score_checksum": sw_sum},
        "sw_int32": dict(i32_leg, metric="config 2 through the int32 kernel (AGX_SW_KERNEL_INT32: 32-bit state, one pair at a time per lane group, DNA-coded match: 7.5 instructions per cell)", unit="GCUPS",
                         scores_identical_to_packed=i32_same, useful_cell_fraction=i32_info.cells / max(1, i32_info.padded_cells),
                         roofline=roof(sw.algorithmic_bytes(), i32_t["launch_ms"], "sw_fill_int32", i32_t["back_to_back_launch_ms"])),
        "pairhmm": dict(ph_leg, metric="PairHMM forward pairs/s (config 3: 65536 pairs R=100 H=300 fp32 per GPU)", unit="pairs/s",
                        dtype="f32 (two haplotypes per lane group, packed FMA, two reads per group behind one another (read trains); double rescue)",
                        gcups=n_gpus * ph.cells() * ph_t["steps"] / ph_t["dt"] / 1e9, rescued_in_f64=int(ph_rescued),
                        waves=ph_info.n_waves, launches_per_step=ph_info.n_launches,
                        useful_cell_fraction=ph_info.cells / max(1, ph_info.padded_cells),
                        roofline=roof(ph.algorithmic_bytes(), ph_t["launch_ms"], "phmm_fill", ph_t["back_to_back_launch_ms"]),
                        valu={"ops_per_cell": "9 instructions per 2 cells (fast cell): v_pk_fma_f32 x4, v_pk_mul_f32 x2, v_pk_add_f32, v_perm_b32 x2",
                              "frac_of_instruction_mix_ceiling": ph_info.padded_cells * (9 / 2 / VALU_PACKED) / (ph_t["back_to_back_launch_ms"] * 1e-3),
                              "priced_on": "kernel_only.launch_ms (the fill with its results left in HBM)"},
                        log10_checksum=ph_sum),
    }
    if extra:
        c4_leg, c5_leg = leg(c4_t, c4.cells(sentinel=False) / 1e9, "gcups"), leg(c5_t, c5.n_pairs, "pairs_per_s")

        def strong(t, rows, total_units, unit_name, cells_key):
            cells = np.array([r["cells"] for r in rows], dtype=np.float64)
            return {"value": total_units * t["steps"] / t["dt"], "unit": unit_name, "steps": t["steps"], "ms_per_step": t["dt"] / t["steps"] * 1e3,
                    "window": WINDOW, "scaling": "strong", "n_shards": len(rows),
                    "shard_cell_imbalance": float(cells.max() / cells.mean()) if cells.size and cells.mean() > 0 else None,
                    "launch_ms_min_over_ranks": min(r["launch_ms"] for r in rows), "launch_ms_max_over_ranks": max(r["launch_ms"] for r in rows)}

        out["config4"] = dict(c4_leg, metric="Smith-Waterman GCUPS, mixed lengths 32-512 (config 4: 1 048 576 pairs over 8 GPUs; weak leg: %d pairs per GPU)" % (C4_PAIRS // 8),
                              unit="GCUPS", launches_per_step=c4_info.n_launches, useful_cell_fraction=c4_info.cells / max(1, c4_info.padded_cells),
                              score_checksum=c4_sum, roofline=roof(c4.algorithmic_bytes(), c4_t["launch_ms"], "sw_fill_c4shard"),
                              total=dict(strong(s4_t, s4_rows, c4f.cells(sentinel=False) / 1e9, "GCUPS", "cells"),
                                         workload="ONE batch of 1 048 576 pairs U[32,512] cut into %d shards by cells (agx_sw_shard_cuts)" % world,
                                         launches_per_step_rank0=s4_info.n_launches, useful_cell_fraction_rank0=s4_info.cells / max(1, s4_info.padded_cells)),
                              per_rank=s4_rows)
        out["config5"] = dict(c5_leg, metric="PairHMM forward pairs/s, fp64 in the reference's operation order (config 5: 262 144 pairs R=250 H=500 over 8 GPUs; weak leg: %d pairs per GPU)" % c5.n_pairs,
                              unit="pairs/s", dtype="f64", gcups=n_gpus * c5.cells() * c5_t["steps"] / c5_t["dt"] / 1e9,
                              launches_per_step=c5_info.n_launches, useful_cell_fraction=c5_info.cells / max(1, c5_info.padded_cells),
                              log10_checksum=c5_sum, roofline=roof(c5.algorithmic_bytes(), c5_t["launch_ms"], "phmm_fill_c5shard"),
                              valu={"ops_per_cell": "12 instructions per cell: the reference's 11 fp64 operations in its order (7 v_mul_f64, 4 v_add_f64) + "
                                                    "v_add_u32_sdwa (the prior is read from the read's LDS table, ds_read_b64)",
                                    "frac_of_instruction_mix_ceiling": c5_info.padded_cells * (12 / VALU_PACKED) / (c5_t["launch_ms"] * 1e-3)},
                              total=dict(strong(s5_t, s5_rows, c5f.n_pairs, "pairs/s", "cells"),
                                         workload="ONE batch of 262 144 pairs R=250 H=500 (512 regions) cut into %d shards of whole regions by cells (agx_phmm_shard_cuts)" % world),
                              per_rank=s5_rows,
                              fp64_fma=dict(leg(c5m_t, c5.n_pairs, "pairs_per_s"), unit="pairs/s",
                                            what="the weak leg's shard with explicit fma (AGX_PHMM_F64_FMA): 8 instead of 11 fp64 operations per cell",
                                            max_rel_diff_to_bit_identical=c5m_rel, tolerance_asked=1e-12))
        out["corpus_10s"] = corpus_legs
        out["multi_one_process"] = multi_one
        out["one_shot"] = one
    # the figures of the other legs that a reader of the first few keys should not have to dig for
    summ = {"int32_gcups": i32_leg["value"], "int32_ms_per_step": i32_leg["ms_per_step"], "int32_scores_identical_to_packed": i32_same,
            "pairhmm_config3_f32_pairs_per_s": ph_leg["value"], "pairhmm_config3_ms_per_step": ph_leg["ms_per_step"]}
    if extra:
        summ.update({"config4_shard_gcups": out["config4"]["value"], "config4_total_gcups": out["config4"]["total"]["value"],
                     "config4_total_n_shards": out["config4"]["total"]["n_shards"],
                     "config5_shard_f64_pairs_per_s": out["config5"]["value"], "config5_total_f64_pairs_per_s": out["config5"]["total"]["value"],
                     "multi_one_process_config4_ms": multi_one.get("config4", {}).get("ms"), "multi_one_process_config5_ms": multi_one.get("config5", {}).get("ms"),
                     "multi_one_process_config4_page_locked_ms": multi_one.get("config4_page_locked_source", {}).get("ms"),
                     "one_shot_config2_pinned_ms_min": one["config2_pinned"]["ms_min"], "one_shot_config3_ms_min": one["config3_pageable"]["ms_min"],
                     "corpus_10s_f64_pairs_per_s": corpus_legs.get("f64", {}).get("pairs_per_s"),
                     "corpus_10s_f32_fma_pairs_per_s": corpus_legs.get("f32_fma", {}).get("pairs_per_s"),
                     "corpus_10s_f64_useful_cell_fraction": corpus_legs.get("f64", {}).get("useful_cell_fraction"),
                     "corpus_10s_f32_fma_useful_cell_fraction": corpus_legs.get("f32_fma", {}).get("useful_cell_fraction")})
    out["config"]["int32_gcups"], out["config"]["int32_ms_per_step"] = i32_leg["value"], i32_leg["ms_per_step"]
    out["config"]["other_legs"] = summ
    out["control_plane"] = control
    if not args.no_cpu_baseline and n_gpus == 1:  # the CPU baseline is reported at N=1 only
        cores = host_cores()
        n_par = max(1, min(cores, 64))
        out["cpu_baseline"] = _sw_ref_or_port(synth.sw_pairs(16384, SW_LEN, SW_LEN, seed=2, related_frac=0.25), "16384 of the 65536 config-2 pairs")
        out["cpu_baseline"]["host_cores_available"] = cores
        out["cpu_baseline"]["host_cores_of_the_machine"] = os.cpu_count()
        out["pairhmm"]["cpu_baseline"] = _phmm_ref_or_port(synth.phmm_regions(8, PH_READS, PH_HAPS, PH_R, PH_H, seed=3), "8192 of the 65536 config-3 pairs")
        out["pairhmm"]["cpu_baseline_antidiag"] = _phmm_ref_or_port(synth.phmm_regions(2, 16, 16, PH_R, PH_H, seed=3), "512 config-3-shaped pairs (2 regions of 16 reads x 16 haplotypes)",
                                                                    program="phmm_antidiag_ref")
        ph_multi = cpu_baseline_phmm_multicore(n_par, 2)
        if ph_multi:
            out["pairhmm"]["cpu_baseline_multicore"] = ph_multi
        multi_cpu = cpu_baseline_sw_multicore(n_par, 4096)
        if multi_cpu:
            out["cpu_baseline_multicore"] = multi_cpu
        # the driver's record keeps "cpu_baseline" whole: the other baselines are repeated inside it
        out["cpu_baseline"]["pairhmm_config3"] = out["pairhmm"]["cpu_baseline"]
        out["cpu_baseline"]["pairhmm_config3_antidiag"] = out["pairhmm"]["cpu_baseline_antidiag"]
        out["cpu_baseline"]["pairhmm_config3_multicore"] = ph_multi
        out["cpu_baseline"]["sw_multicore"] = multi_cpu
        if extra:
            out["config4"]["cpu_baseline"] = _sw_ref_or_port(synth.sw_pairs(4096, 32, 512, seed=4), "4096 config-4 pairs (lengths U[32,512])")
            out["config5"]["cpu_baseline"] = _phmm_ref_or_port(synth.phmm_regions(8, C5_READS, C5_HAPS, C5_R, C5_H, seed=5), "4096 config-5 pairs (R=250 H=500)")
    print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
