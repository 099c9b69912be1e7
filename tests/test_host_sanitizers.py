"""AddressSanitizer + UBSan, and ThreadSanitizer, over the host side of libagx (text readers, planners, packers) on the CPU
build: the device objects are linked unchanged, no device is touched (plan-only batches)."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "accelerating-genomics_amd")
CLANG = "/opt/rocm/lib/llvm/bin/clang"


def _run_driver(tmp_path, golden_dir, sanitizers, env_extra):
    import accelerating_genomics_amd.api as agx

    if not os.path.exists(os.path.join(PKG, "build", "agx_phmm_finish_kernel.o")):
        agx.build()
    san = ["-fsanitize=" + sanitizers, "-fno-omit-frame-pointer", "-g", "-O1"]
    inc = ["-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-DAGX_TUNING"]  # AGX_HOST_THREADS below
    objs = []
    for src in ("agx_runtime.cpp", "agx_sw.cpp", "agx_phmm.cpp"):
        o = str(tmp_path / (src + ".o"))
        subprocess.run([CLANG + "++", "-std=c++17", "-x", "c++", *san, *inc, "-c", os.path.join(PKG, "csrc", src), "-o", o], check=True)
        objs.append(o)
    for src, extra in ((os.path.join(PKG, "csrc", "agx_text.c"), ["-D_POSIX_C_SOURCE=200809L"]),
                       (os.path.join(ROOT, "tests", "host", "sanitize_driver.c"), [])):
        o = str(tmp_path / (os.path.basename(src) + ".o"))
        subprocess.run([CLANG, "-std=c99", *san, *inc, *extra, "-c", src, "-o", o], check=True)
        objs.append(o)
    dev = sorted(glob.glob(os.path.join(PKG, "build", "agx_*_kernel.o")))  # every device object, linked unchanged
    exe = str(tmp_path / "sanitize_driver")
    subprocess.run([CLANG + "++", *san, *objs, *dev, "-o", exe, "-L/opt/rocm/lib", "-lamdhip64", "-lpthread",
                    "-Wl,-rpath,/opt/rocm/lib"], check=True)
    env = dict(os.environ, AGX_HOST_THREADS="4", **env_extra)
    r = subprocess.run([exe, golden_dir], capture_output=True, env=env, timeout=600)
    tail = (r.stdout + r.stderr).decode(errors="replace")[-3000:]
    assert r.returncode == 0 and b"SANITIZE_DRIVER_OK" in r.stdout, tail
    assert b"runtime error" not in r.stderr and b"Sanitizer" not in r.stderr, tail


@pytest.mark.skipif(not os.path.exists(CLANG), reason="ROCm clang not present")
def test_host_code_under_asan_ubsan(tmp_path, golden_dir):
    _run_driver(tmp_path, golden_dir, "address,undefined",
                dict(ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"))


@pytest.mark.skipif(not os.path.exists(CLANG), reason="ROCm clang not present")
def test_host_code_under_tsan(tmp_path, golden_dir):
    """The threaded planners and readers (four pool threads) under ThreadSanitizer: no data race."""
    _run_driver(tmp_path, golden_dir, "thread", dict(TSAN_OPTIONS="halt_on_error=1"))


def test_fast_f6_formatter_is_printf(tmp_path):
    """host/agx_fmt.h writes "%f\\n" for the PairHMM command line: byte for byte snprintf's output on special values,
    every multiple of 2^-7 and 2^-12 around zero and its neighbours (exact ties, near-ties), likelihood-like values,
    decimal near-ties, large values and random bit patterns (tests/host/fmt_check.c; 2.4 million values here)."""
    exe = str(tmp_path / "fmt_check")
    subprocess.run(["gcc", "-O2", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-D_DEFAULT_SOURCE", "-fsanitize=undefined", "-fno-sanitize-recover=undefined",
                    "-I" + os.path.join(PKG, "host"), os.path.join(ROOT, "tests", "host", "fmt_check.c"), "-o", exe, "-lm"], check=True)
    r = subprocess.run([exe, "300000"], capture_output=True, timeout=600)
    assert r.returncode == 0 and b"FMT_CHECK_OK" in r.stdout, (r.stdout + r.stderr).decode(errors="replace")[-2000:]
