"""A short run of tools/fuzz_gpu.py: random batch structures (empty sides, odd haplotype counts, long
haplotypes, mixed shapes) through every precision and scoring mode against the oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_randomised_parity_sweep():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_gpu.py"), "12", "11"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and "FUZZ_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
