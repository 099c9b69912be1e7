"""bench.py's own rank launcher: `python bench.py --gpus N` run bare must start N ranks
before anything touches HIP, relay rank 0's line and fail when a rank fails.  No GPU here: every rank ends with
the product's loud "no HIP device" and the launcher must report exactly that."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*argv, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=300, env=e, cwd=ROOT)


def test_bare_gpus_2_starts_two_ranks_before_any_hip_call():
    r = run_bench("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0                      # no GPU in this container: both ranks must fail, and so must the launcher
    assert r.stdout.strip() == ""                 # no JSON line was invented
    assert "starting 2 ranks (parent has imported torch: False, libagx: False, HIP libraries mapped: False)" in r.stderr
    assert "rank 0 of 2: no HIP device visible" in r.stderr and "rank 1 of 2: no HIP device visible" in r.stderr
    assert "rank return codes [1, 1]" in r.stderr


def test_world_size_mismatch_is_an_error_not_a_silent_n_gpus_1():
    r = run_bench("--gpus", "8", "--steps", "1", "--warmup", "0", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "start as many ranks as GPUs asked for" in r.stderr and r.stdout.strip() == ""


def test_launcher_source_order():
    """The launcher branch sits in main() before the first torch / libagx import."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index("self_launch(args)") < main.index("import torch") < main.index("agx.device_count()")
    head = src[: src.index("def main():")]
    for line in head.splitlines():
        if not line.startswith((" ", "\t")):           # module level only
            assert not line.startswith(("import torch", "from torch", "import accelerating_genomics_amd")), line
    assert json  # keep the import used
