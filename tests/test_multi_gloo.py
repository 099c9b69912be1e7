"""N > 1 path on CPU: world_size-2 gloo.  Each rank takes its shard of one host batch by the
same cut rule the C library uses, scores it (the oracle stands in for the device in this CPU
test), and rank 0 reassembles the slices; the max-over-ranks clock reduction is exercised too."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

import accelerating_genomics_amd.dist as agd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_are_contiguous_balanced_and_complete():
    rng = np.random.default_rng(0)
    cost = rng.integers(1, 500, size=1000) ** 2
    for world in (1, 2, 3, 8):
        cut = agd.shard_bounds(cost, world)
        assert cut[0] == 0 and cut[-1] == cost.size and np.all(np.diff(cut) >= 0)
        loads = [cost[cut[r] : cut[r + 1]].sum() for r in range(world)]
        assert max(loads) <= cost.sum() / world + cost.max() + world
    assert list(agd.shard_bounds([], 4)) == [0, 0, 0, 0, 0]
    assert list(agd.shard_bounds([5], 4)) == [0, 1, 1, 1, 1]


WORKER = textwrap.dedent("""
    import os, sys, time, numpy as np
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    import accelerating_genomics_amd.dist as agd, accelerating_genomics_amd.synth as synth
    from tests import oracle_api
    rank, local, world = agd.env_rank()
    dist.init_process_group("gloo")
    orc = oracle_api.load()
    # Smith-Waterman: pairs sharded by cells
    b = synth.sw_pairs(301, 10, 120, seed=5, related_frac=0.5)
    l = b.len.astype(np.int64)
    cut = agd.shard_bounds(l[0::2] * l[1::2], world)
    mine = b.subset(np.arange(cut[rank], cut[rank + 1]))
    dist.barrier()
    t0 = time.perf_counter()
    local_scores = orc.sw_batch(mine)
    dt = time.perf_counter() - t0 + rank  # rank 1 is "slower" by construction
    slow = agd.max_over_ranks(dt)
    assert slow >= 1.0, slow
    full = agd.gather_slices(local_scores, cut)
    # PairHMM: whole regions sharded
    p = synth.phmm_regions(5, 3, 2, 30, 50, seed=6, jitter=5)
    R, H = p.pair_lengths()
    per_region = [int(x) for x in np.add.reduceat(R * H, np.cumsum([0] + [6] * 4))]
    rcut = agd.shard_bounds(per_region, world)
    sub = p.regions(int(rcut[rank]), int(rcut[rank + 1]))
    _, ll = orc.phmm_batch(sub, 0)
    pcut = np.array([6 * int(c) for c in rcut])
    pfull = agd.gather_slices(ll, pcut)
    if rank == 0:
        assert np.array_equal(full.astype(np.int32), orc.sw_batch(b))
        assert np.array_equal(pfull, orc.phmm_batch(p, 0)[1])
        print("GLOO_OK", world, flush=True)
    dist.barrier()
    dist.destroy_process_group()
""")


def test_world_size_2_gloo_shards_and_reassembles():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER % ROOT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, cwd=ROOT))
    outs = [p.communicate(timeout=180) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e.decode()[-2000:]
    assert b"GLOO_OK 2" in outs[0][0]
