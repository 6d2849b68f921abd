"""The C-ABI library loads on a CPU-only box and exports exactly what
include/thzgpu.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

import thz_image_explorer_amd as pkg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "thzgpu.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(thz_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    lib = pkg.load_library()
    names = _declared()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f"{n} declared in thzgpu.h but not exported by libthzgpu.so"


def test_binding_covers_header():
    bound = {s[0] for s in pkg.SYMBOLS}
    assert set(_declared()) == bound


def test_abi_version():
    assert pkg.load_library().thz_abi_version() == 2


def test_no_cpu_fallback():
    """Without a GPU the engine refuses to start instead of computing on the CPU."""
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU present")
    with pytest.raises(pkg.ThzError) as e:
        pkg.Engine(0)
    assert e.value.code == -3


def test_library_does_not_link_oracle():
    out = os.popen(f"ldd {pkg.LIB_PATH}").read()
    assert "thz_oracle" not in out
