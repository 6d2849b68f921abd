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


RUST = os.path.join(ROOT, "rust")


def _rust_sources():
    out = {}
    for root, _, files in os.walk(RUST):
        for f in files:
            if f.endswith((".rs", ".patch")):
                out[os.path.relpath(os.path.join(root, f), RUST)] = open(os.path.join(root, f)).read()
    return out


def test_rust_binding_calls_only_what_exists():
    """No Rust toolchain here, so what a compiler would catch first is checked by text: every engine method the
    plugins / math_tools_gpu / the data-thread patch call exists in engine.rs, every math_tools_gpu function they
    call exists there, and engine.rs is the transliteration of the C++ engine that IS built and tested
    (thz_image_explorer_amd/host/thz_engine.hpp: same method names)"""
    src = _rust_sources()
    engine_fns = set(re.findall(r"pub fn (\w+)\s*[(<]", src["engine.rs"]))
    mt_fns = set(re.findall(r"pub fn (\w+)\s*[(<]", src["math_tools_gpu.rs"]))
    for name, text in src.items():
        if name in ("ffi.rs", "engine.rs"):
            continue
        code = "\n".join(l for l in text.splitlines() if not l.lstrip().startswith(("//", "+ //", "+//")))
        used = set(re.findall(r"\beng\.(\w+)\(", code)) | set(re.findall(r"ENGINE\.lock\(\)\.unwrap\(\)\.(\w+)\(", code))
        assert used <= engine_fns, f"{name}: engine methods that do not exist: {sorted(used - engine_fns)}"
        used_mt = set(re.findall(r"math_tools_gpu::(\w+)\(", code))
        assert used_mt <= mt_fns, f"{name}: math_tools_gpu functions that do not exist: {sorted(used_mt - mt_fns)}"
        for imp in re.findall(r"use crate::math_tools_gpu::\{([^}]*)\}", text) + re.findall(r"use crate::math_tools_gpu::(\w+);", text):
            for fn in [x.strip() for x in imp.split(",") if x.strip()]:
                assert fn in mt_fns, f"{name} imports math_tools_gpu::{fn}"
    hpp = open(os.path.join(ROOT, "thz_image_explorer_amd", "host", "thz_engine.hpp")).read()
    cpp_methods = set(re.findall(r"\b(\w+)\s*\(", hpp))
    twin = engine_fns - {"new", "chain_position_of_domain", "chain_position_of_id", "voxels"}
    assert twin <= cpp_methods, f"engine.rs methods without a C++ twin: {sorted(twin - cpp_methods)}"
    # no plugin file refers to UI modules that exist nowhere
    for name, text in src.items():
        assert "band_pass_fd_ui" not in text and "band_pass_td_ui" not in text, name


def test_data_thread_patch_applies_to_the_reference():
    ref = "/root/reference/src/data_thread.rs"
    if not os.path.exists(ref):
        pytest.skip("the reference is not on this machine")
    import shutil
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "src"))
        shutil.copy(ref, os.path.join(d, "src", "data_thread.rs"))
        r = subprocess.run(["patch", "-p1", "--dry-run", "-i", os.path.join(RUST, "data_thread.patch")], cwd=d,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0 and "FAILED" not in r.stdout and "fuzz" not in r.stdout, r.stdout
