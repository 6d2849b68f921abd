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


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs clang with the ThreadSanitizer runtime")
def test_kernels_are_race_free():
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 history_size=2 exitcode=0")
    p = subprocess.run(["bash", os.path.join(HERE, "emu", "run_tsan.sh")], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=1500)
    out = p.stdout
    if "unsupported option '-fsanitize=thread'" in out or "cannot find" in out and "tsan" in out:
        pytest.skip("ThreadSanitizer runtime not available")
    assert p.returncode == 0, out[-4000:]
    for tag in ("chain nt=4096 allow_f=1 done", "chain nt=1001 allow_f=1 done", "chain nt=5000 allow_f=1 done",
                "rl done", "dc done", "helpers done", "voxel done"):
        assert tag in out, f"driver did not reach: {tag}\n{out[-2000:]}"
    reports = [l for l in out.splitlines() if "WARNING: ThreadSanitizer" in l]
    assert not reports, out[-6000:]
