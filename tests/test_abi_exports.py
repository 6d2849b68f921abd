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


def _declared_in(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(thz_[a-z0-9_]+)\s*\(", src))


def test_rust_ffi_is_current_and_complete():
    """rust/ffi.rs (the extern "C" block + #[repr(C)] structs a Rust maintainer binds) is generated from the headers:
    it must be what the generator produces now, declare every function of thzgpu.h / thzio.h and nothing else, and
    carry every struct of the headers with the same field count"""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gen_rust_ffi.py"), "--check"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    ffi = open(os.path.join(ROOT, "rust", "ffi.rs")).read()
    fns = set(re.findall(r"pub fn (thz_[a-z0-9_]+)\(", ffi))
    assert fns == _declared_in("thzgpu.h") | _declared_in("thzio.h")
    for header in ("thzgpu.h", "thzio.h"):
        src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", header)).read(), flags=re.S)
        for body, name in re.findall(r"typedef\s+struct\s*\w*\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
            camel = "".join(p.capitalize() for p in name.split("_"))
            m = re.search(r"pub struct %s \{(.*?)\n\}" % camel, ffi, flags=re.S)
            assert m, f"{name} missing from rust/ffi.rs"
            n_c = sum(len(stmt.split(",")) for stmt in body.split(";") if stmt.strip())
            assert len(re.findall(r"pub \w+:", m.group(1))) == n_c, name
    # the hand-written shims name only functions that exist
    for root, _, files in os.walk(os.path.join(ROOT, "rust")):
        for f in files:
            if f.endswith(".rs") and f != "ffi.rs":
                used = set(re.findall(r"\b(thz_[a-z0-9_]+)\s*\(", open(os.path.join(root, f)).read()))
                assert used <= fns, f"{f}: {sorted(used - fns)}"
