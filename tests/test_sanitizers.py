"""Race detection for the HIP kernels on the CPU: the emulation (every lane a host thread, barriers as
pthread barriers, tests/emu/) built with ThreadSanitizer and driven over every kernel family by
tests/emu/tsan_driver.cpp.  A report is a pair of LDS / global accesses of two lanes that no barrier orders,
i.e. code that only works while a wave happens to run in lock-step."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def _run(san, families=()):
    env = dict(os.environ, SAN=san, TSAN_OPTIONS="halt_on_error=0 history_size=2 exitcode=0",
               UBSAN_OPTIONS="print_stacktrace=1")
    return subprocess.run(["bash", os.path.join(HERE, "emu", "run_tsan.sh"), *families], env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, text=True, timeout=1500)


REACHED = ("chain nt=4096 allow_f=1 done", "chain nt=1001 allow_f=1 done", "chain nt=5000 allow_f=1 done",
           "sums nt=4096 cmask=1", "sums nt=1001 cmask=1",   # the in-launch pixel sums: lock-free ticket hand-over in LDS
           "rl done", "dc done", "helpers done", "voxel done")


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs clang with the sanitizer runtimes")
def test_kernels_are_race_free():
    p = _run("thread")
    out = p.stdout
    if "unsupported option '-fsanitize=thread'" in out or "cannot find" in out and "tsan" in out:
        pytest.skip("ThreadSanitizer runtime not available")
    assert p.returncode == 0, out[-4000:]
    for tag in REACHED:
        assert tag in out, f"driver did not reach: {tag}\n{out[-2000:]}"
    reports = [l for l in out.splitlines() if "WARNING: ThreadSanitizer" in l]
    assert not reports, out[-6000:]


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs clang with the sanitizer runtimes")
@pytest.mark.skipif(os.environ.get("THZ_SANITIZE_ALL") != "1",
                    reason="two more minutes of build + run; set THZ_SANITIZE_ALL=1 (last run clean: round 2, after the P-kernel and RL-kernel changes)")
def test_kernels_stay_in_bounds():
    """the same driver under AddressSanitizer + UBSan: the emulation's LDS is a heap block of exactly the
    launch's dynamic-LDS size, so a lane reading or writing past it (the GPU would hand back zeros or a
    neighbour's data) is an error here, as is any global access outside the caller's arrays"""
    # the cooperative chirp-z kernels (nt = 3000, 5000) take minutes under this sanitizer: the other families
    p = _run("address,undefined", ("f", "sums", "g", "fb", "rl", "dc", "helpers", "voxel"))
    out = p.stdout
    if "unsupported option" in out:
        pytest.skip("sanitizer runtime not available")
    assert p.returncode == 0, out[-4000:]
    for tag in REACHED:
        if "nt=5000" not in tag:
            assert tag in out, f"driver did not reach: {tag}\n{out[-2000:]}"
    assert "AddressSanitizer" not in out and "runtime error" not in out, out[-6000:]


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs clang with the sanitizer runtimes")
def test_host_math_under_sanitizers(tmp_path):
    """the O(Nt) host math (windows, band-pass index rules, tilt plan, reference alignment, FIR bank, PSF
    evaluation) under AddressSanitizer + UBSan, on ordinary and on edge inputs (tests/emu/host_san_driver.cpp)"""
    csrc = os.path.join(HERE, "..", "thz_image_explorer_amd", "csrc")
    exe = str(tmp_path / "host_san")
    b = subprocess.run([CLANG, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined",
                        "-fno-sanitize=float-divide-by-zero", "-ffp-contract=off", "-I" + csrc,
                        "-I" + os.path.join(HERE, "..", "include"), os.path.join(HERE, "emu", "host_san_driver.cpp"),
                        os.path.join(csrc, "host_windows.cpp"), os.path.join(csrc, "deconv_host.cpp"), "-lm", "-o", exe],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if b.returncode != 0 and "unsupported option" in b.stdout:
        pytest.skip("sanitizer runtime not available")
    assert b.returncode == 0, b.stdout[-3000:]
    r = subprocess.run([exe], env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1"),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "host math done" in r.stdout, r.stdout[-4000:]
    assert "runtime error" not in r.stdout and "AddressSanitizer" not in r.stdout, r.stdout[-4000:]


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs clang with the sanitizer runtimes")
@pytest.mark.skipif(not os.path.exists("/opt/conda/include/hdf5.h"), reason="HDF5 headers not installed")
def test_thz_io_under_sanitizers(tmp_path):
    """the dotTHz reader / writer under AddressSanitizer + UBSan: a real sample file, write + re-read, caller
    buffers smaller than the values, a missing file and a file that is not HDF5 (tests/emu/io_san_driver.cpp)"""
    root = os.path.join(HERE, "..")
    exe = str(tmp_path / "io_san")
    b = subprocess.run([CLANG, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-I" + os.path.join(root, "include"),
                        "-I/opt/conda/include", os.path.join(HERE, "emu", "io_san_driver.cpp"),
                        os.path.join(root, "thz_image_explorer_amd", "io", "thz_io.cpp"), "/opt/conda/lib/libhdf5.so",
                        "-o", exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if b.returncode != 0 and "unsupported option" in b.stdout:
        pytest.skip("sanitizer runtime not available")
    assert b.returncode == 0, b.stdout[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1",
               LD_LIBRARY_PATH="/usr/lib/x86_64-linux-gnu:/opt/conda/lib")   # the system's libstdc++, conda's HDF5
    r = subprocess.run([exe, os.path.join(HERE, "golden", "knife_edge_2groups.thz"), str(tmp_path)], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "io done" in r.stdout, r.stdout[-4000:]
    assert "runtime error" not in r.stdout and "AddressSanitizer" not in r.stdout, r.stdout[-4000:]
