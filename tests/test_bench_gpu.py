"""bench.py's rank plumbing on the one-GPU box: the nccl (= RCCL) control plane at world size 1 --
init_process_group(device_id=...), all_reduce(MAX) and all_gather on device tensors execute once on hardware -- and
the bare `--gpus 2` self-launch with both ranks on device 0 (a rehearsal of the control flow, never a number)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QUICK = ["--steps", "6", "--warmup", "2", "--warm-seconds", "0.05", "--no-cpu-baseline"]


def clean_env(**extra):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e.update(extra)
    return e


def json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_nccl_control_plane_at_world_size_1():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = clean_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dist-backend", "nccl", "--no-extra-configs", *QUICK],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json_line(r.stdout)
    assert out["n_gpus"] == 1 and out["control_plane"]["backend"] == "nccl" and out["control_plane"]["tensors_on"] == "cuda"
    assert out["control_plane"]["world_size"] == 1 and out["value"] > 100.0


def test_bare_gpus_2_on_one_device_runs_two_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-device", *QUICK],
                       capture_output=True, text=True, timeout=1100, env=clean_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "starting 2 ranks (parent has imported torch: False, libagx: False, HIP libraries mapped: False)" in r.stderr
    out = json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["control_plane"] == dict(out["control_plane"], backend="gloo", world_size=2, self_launched=True)
    assert out["config4"]["total"]["n_shards"] == 2 and len(out["config4"]["per_rank"]) == 2 and len(out["config5"]["per_rank"]) == 2
    assert out["config"]["int32_gcups"] > 100.0 and "cpu_baseline" not in out   # the CPU baseline is an N=1 leg


def test_ranks_that_would_share_a_gpu_are_refused_without_the_rehearsal_flag():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", *QUICK, "--no-extra-configs"],
                       capture_output=True, text=True, timeout=600, env=clean_env(), cwd=ROOT)
    n_dev = int(subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True).stdout.strip() or 0)
    if n_dev >= 2:
        assert r.returncode == 0
    else:
        assert r.returncode != 0 and "is not a 2-GPU number" in r.stderr and r.stdout.strip() == "", r.stderr[-2000:]
