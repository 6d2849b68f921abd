"""dotTHz reader / writer (libthzio.so, include/thzio.h) against the reference's loader rules
(open_scan_from_thz io.rs:496-631, open_pulse_from_thz io.rs:435-477) on a real sample file
of the reference (two groups copied with h5copy, tests/golden/knife_edge_2groups.thz) and on
files written by thz_io_save_scan."""
import ctypes
import os
import re

import numpy as np
import pytest

from thz_image_explorer_amd import io_binding as tio

pytestmark = pytest.mark.skipif(not tio.available(), reason="libthzio.so not built (no hdf5.h)")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REAL = os.path.join(GOLD, "knife_edge_2groups.thz")


def test_header_symbols_exported_and_bound():
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "thzio.h")).read(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(thz_io_[a-z0-9_]+)\s*\(", src)))
    lib = ctypes.CDLL(tio.LIB_PATH)
    for n in declared:
        assert hasattr(lib, n), n
    assert set(declared) == {s[0] for s in tio.SYMBOLS}


def test_real_single_pulse_file():
    k = np.load(os.path.join(GOLD, "knife_edge.npz"))
    with tio.ScanFile(REAL) as f:
        assert f.group_count == 2
        assert f.group_name == "Beam Width Measurement x=-0.10"      # first group in name order
        assert (f.nx, f.ny, f.nt, f.kind) == (1, 1, 1001, 1)          # "single pulse dataset", io.rs:548-562
        t, c = f.time(), f.cube()
        assert c.shape == (1, 1, 1001)
        i = list(k["groups"]).index("/Beam Width Measurement x=-0.10/ds1")
        assert np.array_equal(t, k["time"]) and np.array_equal(c[0, 0], k["traces"][i])
        # metadata map: mdDescription "position [mm], axis" -> md1 (f64 -0.1), md2 ("x")
        assert f.metadata("position [mm]") == repr(-0.09999999999999787) and f.metadata("axis") == "x"
        assert f.metadata("width") is None
        assert f.attribute("thzVer") == "1.00" and f.attribute("dsDescription") == "x"
        g = f.geometry()
        assert (g.width, g.height, g.dx, g.dy, g.has_dx, g.has_x_min) == (1, 1, 1.0, 1.0, 1, 0)
    t2, s2 = tio.read_pulse(REAL)                                     # open_pulse_from_thz
    assert np.array_equal(t2, t) and np.array_equal(s2, c[0, 0])


def test_scan_round_trip_slabs_and_geometry(tmp_path):
    rng = np.random.default_rng(2)
    nx, ny, nt = 7, 5, 64
    time = (1000 + 0.05 * np.arange(nt)).astype(np.float32)
    cube = rng.standard_normal((nx, ny, nt)).astype(np.float32)
    path = str(tmp_path / "scan.thzimg")
    md = {"width": nx, "height": ny, "dx [mm]": "0.5", "dy [mm]": "0.25", "x_min [mm]": "-3", "y_min [mm]": "oops",
          "comment": "a, b"}
    tio.save_scan(path, time, cube, md)
    with tio.ScanFile(path) as f:
        assert (f.nx, f.ny, f.nt, f.kind) == (nx, ny, nt, 0) and f.group_name == "Image"
        assert np.array_equal(f.time(), time)
        assert np.array_equal(f.cube(), cube)
        parts = [f.cube(0, 3), f.cube(3, 1), f.cube(4, 3)]             # streamed x-slabs
        assert np.array_equal(np.concatenate(parts), cube)
        with pytest.raises(tio.ThzIoError):
            f.cube(5, 3)
        g = f.geometry()
        assert (g.width, g.height, g.dx, g.dy, g.x_min) == (nx, ny, 0.5, 0.25, -3.0)
        assert g.has_y_min == 0                                        # "oops".parse::<f32>() fails -> None
        assert f.attribute("dsDescription") == "time, dataset"
    # a scan file is not a pulse file: first dataset is 1-D -> empty vectors (io.rs:466-471)
    t, s = tio.read_pulse(path)
    assert t.size == 0 and s.size == 0


def test_metadata_overrides_and_unparsable_width(tmp_path):
    time = np.arange(8, dtype=np.float32)
    cube = np.zeros((4, 3, 8), np.float32)
    path = str(tmp_path / "m.thz")
    tio.save_scan(path, time, cube, {"width": "2", "height": "3.0"})    # "3.0".parse::<usize>() fails
    with tio.ScanFile(path) as f:
        g = f.geometry()
        assert (g.width, g.height) == (2, 3)
        assert (g.has_dx, g.has_dy) == (0, 0)


def test_pulse_round_trip_and_errors(tmp_path):
    t = np.linspace(0, 10, 101).astype(np.float32)
    s = np.sin(t).astype(np.float32)
    path = str(tmp_path / "ref.thz")
    tio.save_pulse(path, "Reference", t, s)
    t2, s2 = tio.read_pulse(path)
    assert np.array_equal(t2, t) and np.array_equal(s2, s)
    with tio.ScanFile(path) as f:
        assert f.kind == 1 and np.array_equal(f.cube()[0, 0], s)
    with pytest.raises(tio.ThzIoError) as e:
        tio.ScanFile(str(tmp_path / "missing.thz"))
    assert e.value.code == -2
    bad = tmp_path / "lfs_pointer.thzimg"                              # what the LFS stubs in sample_data look like
    bad.write_text("version https://git-lfs.github.com/spec/v1\n")
    with pytest.raises(tio.ThzIoError):
        tio.ScanFile(str(bad))
