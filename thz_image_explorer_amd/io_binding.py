"""ctypes binding of libthzio.so (include/thzio.h): dotTHz files for tests and tools."""
import ctypes as C
import os

import numpy as np

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libthzio.so")
_P = C.c_void_p
_SZ = C.c_size_t


class Geometry(C.Structure):
    _fields_ = [("width", _SZ), ("height", _SZ), ("dx", C.c_float), ("dy", C.c_float), ("x_min", C.c_float),
                ("y_min", C.c_float), ("has_dx", C.c_int32), ("has_dy", C.c_int32), ("has_x_min", C.c_int32),
                ("has_y_min", C.c_int32)]


SYMBOLS = [
    ("thz_io_last_error", C.c_char_p, []),
    ("thz_io_open", C.c_int, [C.c_char_p, C.POINTER(_P)]),
    ("thz_io_close", None, [_P]),
    ("thz_io_group_count", _SZ, [_P]),
    ("thz_io_group_name", C.c_char_p, [_P]),
    ("thz_io_shape", C.c_int, [_P, C.POINTER(_SZ), C.POINTER(_SZ), C.POINTER(_SZ), C.POINTER(C.c_int)]),
    ("thz_io_read_time", C.c_int, [_P, _P]),
    ("thz_io_read_cube", C.c_int, [_P, _SZ, _SZ, _P]),
    ("thz_io_metadata", C.c_long, [_P, C.c_char_p, C.c_char_p, _SZ]),
    ("thz_io_attribute", C.c_long, [_P, C.c_char_p, C.c_char_p, _SZ]),
    ("thz_io_get_geometry", C.c_int, [_P, C.POINTER(Geometry)]),
    ("thz_io_read_pulse", C.c_int, [C.c_char_p, C.POINTER(_SZ), _P, _P]),
    ("thz_io_save_scan", C.c_int, [C.c_char_p, _P, _SZ, _P, _SZ, _SZ, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), _SZ]),
    ("thz_io_save_pulse", C.c_int, [C.c_char_p, C.c_char_p, _P, _P, _SZ]),
]

_lib = None


class ThzIoError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"thzio error {code}: {msg}")
        self.code = code


def available() -> bool:
    return os.path.exists(LIB_PATH)


def load_library():
    global _lib
    if _lib is None:
        if not available():
            raise ThzIoError(-2, f"{LIB_PATH} is missing: run `make -C thz_image_explorer_amd/io` (needs hdf5.h)")
        lib = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def _check(rc):
    if rc != 0:
        raise ThzIoError(rc, load_library().thz_io_last_error().decode())


class ScanFile:
    """An open .thz / .thzimg file (first group), open_scan_from_thz's view of it."""

    def __init__(self, path):
        self.lib = load_library()
        self.h = _P()
        _check(self.lib.thz_io_open(os.fsencode(path), C.byref(self.h)))
        nx, ny, nt, kind = _SZ(), _SZ(), _SZ(), C.c_int()
        _check(self.lib.thz_io_shape(self.h, C.byref(nx), C.byref(ny), C.byref(nt), C.byref(kind)))
        self.nx, self.ny, self.nt, self.kind = nx.value, ny.value, nt.value, kind.value

    def close(self):
        if self.h:
            self.lib.thz_io_close(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def group_count(self):
        return self.lib.thz_io_group_count(self.h)

    @property
    def group_name(self):
        return self.lib.thz_io_group_name(self.h).decode()

    def time(self):
        t = np.empty(self.nt, np.float32)
        _check(self.lib.thz_io_read_time(self.h, t.ctypes.data))
        return t

    def cube(self, x0=0, n=None):
        n = self.nx - x0 if n is None else n
        out = np.empty((n, self.ny, self.nt), np.float32)
        _check(self.lib.thz_io_read_cube(self.h, x0, n, out.ctypes.data))
        return out

    def _text(self, fn, key):
        n = fn(self.h, key.encode(), None, 0)
        if n < 0:
            return None
        buf = C.create_string_buffer(n + 1)
        fn(self.h, key.encode(), buf, n + 1)
        return buf.value.decode()

    def metadata(self, key):
        return self._text(self.lib.thz_io_metadata, key)

    def attribute(self, name):
        return self._text(self.lib.thz_io_attribute, name)

    def geometry(self) -> Geometry:
        g = Geometry()
        _check(self.lib.thz_io_get_geometry(self.h, C.byref(g)))
        return g


def read_pulse(path):
    lib = load_library()
    n = _SZ()
    _check(lib.thz_io_read_pulse(os.fsencode(path), C.byref(n), None, None))
    t, s = np.empty(n.value, np.float32), np.empty(n.value, np.float32)
    if n.value:
        _check(lib.thz_io_read_pulse(os.fsencode(path), C.byref(n), t.ctypes.data, s.ctypes.data))
    return t, s


def save_scan(path, time, cube, md=None):
    t = np.ascontiguousarray(time, np.float32)
    c = np.ascontiguousarray(cube, np.float32)
    md = md or {}
    keys = (C.c_char_p * len(md))(*[k.encode() for k in md])
    vals = (C.c_char_p * len(md))(*[str(v).encode() for v in md.values()])
    _check(load_library().thz_io_save_scan(os.fsencode(path), t.ctypes.data, t.size, c.ctypes.data, c.shape[0],
                                           c.shape[1], keys, vals, len(md)))


def save_pulse(path, group, time, signal):
    t = np.ascontiguousarray(time, np.float32)
    s = np.ascontiguousarray(signal, np.float32)
    _check(load_library().thz_io_save_pulse(os.fsencode(path), group.encode(), t.ctypes.data, s.ctypes.data, t.size))
