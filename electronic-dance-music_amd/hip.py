"""ctypes mirror of the C ABI in include/edm_hip.h (csrc/libedm_hip.so).

Mirrors the reference's operator interface for the hot path -- GaussGrid
(get_value_deriv, add_value, write, ...) and EDMBias (update_forces, add_hills,
pre/add/post_add_hill, write_bias, ...) -- with the same names and argument
meaning, so parity tests read like the reference's own tests.  Bulk arrays live
in HBM (``DeviceArray``).  There is NO CPU path here: if the HIP library is
missing or no GPU is visible, calls raise.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libedm_hip.so")
HEADER = os.path.join(os.path.dirname(HERE), "include", "edm_hip.h")

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)
vp = C.c_void_p


class EdmHipError(RuntimeError):
    pass


class Geometry(C.Structure):
    _fields_ = [
        ("dim", C.c_int), ("interpolate", C.c_int),
        ("n", C.c_int * 3), ("periodic", C.c_int * 3),
        ("min", C.c_double * 3), ("max", C.c_double * 3), ("dx", C.c_double * 3), ("sigma", C.c_double * 3),
        ("boundary_periodic", C.c_int * 3), ("boundary_min", C.c_double * 3), ("boundary_max", C.c_double * 3),
        ("minisize", C.c_int * 3), ("total", C.c_longlong), ("derivatives", C.c_int),
    ]


_PROTOS = {
    # name: (restype, argtypes)
    "edm_hip_last_error": (C.c_char_p, []),
    "edm_hip_version": (C.c_char_p, []),
    "edm_hip_device_count": (C.c_int, [c_ip]),
    "edm_hip_set_device": (C.c_int, [C.c_int]),
    "edm_hip_device_info": (C.c_int, [C.c_char_p, C.c_size_t, c_ip, C.POINTER(C.c_size_t)]),
    "edm_hip_malloc": (C.c_int, [C.POINTER(vp), C.c_size_t]),
    "edm_hip_free": (C.c_int, [vp]),
    "edm_hip_host_malloc": (C.c_int, [C.POINTER(vp), C.c_size_t]),
    "edm_hip_host_free": (C.c_int, [vp]),
    "edm_hip_memcpy_h2d": (C.c_int, [vp, vp, C.c_size_t]),
    "edm_hip_memcpy_d2h": (C.c_int, [vp, vp, C.c_size_t]),
    "edm_hip_memset": (C.c_int, [vp, C.c_int, C.c_size_t]),
    "edm_hip_device_synchronize": (C.c_int, []),
    "edm_hip_grid_create": (C.c_int, [C.POINTER(vp), C.c_int, c_dp, c_dp, c_dp, c_ip]),
    "edm_hip_grid_create_ex": (C.c_int, [C.POINTER(vp), C.c_int, c_dp, c_dp, c_dp, c_ip, C.c_int, C.c_int]),
    "edm_hip_grid_read": (C.c_int, [C.POINTER(vp), C.c_int, C.c_char_p, C.c_int]),
    "edm_hip_grid_reread": (C.c_int, [vp, C.c_char_p]),
    "edm_hip_grid_set_interpolation": (C.c_int, [vp, C.c_int]),
    "edm_hip_grid_destroy": (C.c_int, [vp]),
    "edm_hip_grid_download_derivs": (C.c_int, [vp, c_dp]),
    "edm_hip_grid_upload_derivs": (C.c_int, [vp, c_dp, c_dp]),
    "edm_hip_grid_get_value_deriv": (C.c_int, [vp, C.c_longlong, vp, C.c_int, vp, vp]),
    "edm_hip_grid_add_grid": (C.c_int, [vp, vp, C.c_double, C.c_double]),
    "edm_hip_grid_add_gauss": (C.c_int, [vp, vp, C.c_double, C.c_double]),
    "edm_hip_grid_geometry": (C.c_int, [vp, C.POINTER(Geometry)]),
    "edm_hip_grid_download": (C.c_int, [vp, c_dp]),
    "edm_hip_grid_upload": (C.c_int, [vp, c_dp]),
    "edm_hip_grid_clear": (C.c_int, [vp]),
    "edm_hip_grid_add_values": (C.c_int, [vp, C.c_longlong, vp, C.c_int, vp, C.c_double]),
    "edm_hip_grid_write": (C.c_int, [vp, C.c_char_p]),
    "edm_hip_grid_multi_write": (C.c_int, [vp, C.c_char_p, c_dp, c_dp, c_ip, C.c_int]),
    "edm_hip_gauss_create": (C.c_int, [C.POINTER(vp), C.c_int, c_dp, c_dp, c_dp, c_ip, C.c_int, c_dp]),
    "edm_hip_gauss_read": (C.c_int, [C.POINTER(vp), C.c_int, C.c_char_p, c_dp]),
    "edm_hip_gauss_reread": (C.c_int, [vp, C.c_char_p]),
    "edm_hip_gauss_set_interpolation": (C.c_int, [vp, C.c_int]),
    "edm_hip_gauss_set_lookup_replica": (C.c_int, [vp, C.c_int]),
    "edm_hip_gauss_lookup_replica_info": (C.c_int, [vp, c_ip, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "edm_hip_gauss_destroy": (C.c_int, [vp]),
    "edm_hip_gauss_set_boundary": (C.c_int, [vp, c_dp, c_dp, c_ip]),
    "edm_hip_gauss_geometry": (C.c_int, [vp, C.POINTER(Geometry)]),
    "edm_hip_gauss_download_tables": (C.c_int, [vp, C.c_int, c_dp, c_dp]),
    "edm_hip_gauss_download": (C.c_int, [vp, c_dp, c_dp]),
    "edm_hip_gauss_upload": (C.c_int, [vp, c_dp, c_dp]),
    "edm_hip_gauss_clear": (C.c_int, [vp]),
    "edm_hip_gauss_device_buffer": (C.c_int, [vp, C.POINTER(vp), c_ip, C.POINTER(C.c_longlong)]),
    "edm_hip_gauss_get_value_deriv": (C.c_int, [vp, C.c_longlong, vp, C.c_int, vp, vp]),
    "edm_hip_gauss_sample_index": (C.c_int, [vp, C.c_longlong, vp, C.c_int, vp]),
    "edm_hip_gauss_remap": (C.c_int, [vp, C.c_longlong, vp, C.c_int, vp]),
    "edm_hip_gauss_update_forces": (C.c_int, [vp, C.c_longlong, vp, C.c_int, vp, C.c_int, vp, C.c_int, c_dp]),
    "edm_hip_gauss_pair_forces": (C.c_int, [vp, C.c_longlong, vp, vp, c_dp]),
    "edm_hip_gauss_profile_enable": (C.c_int, [vp, C.c_int]),
    "edm_hip_gauss_profile_read": (C.c_int, [vp, c_dp, C.POINTER(C.c_longlong), C.c_int]),
    "edm_hip_gauss_add_values": (C.c_int, [vp, C.c_longlong, vp, C.c_int, vp, C.c_double, vp, c_dp]),
    "edm_hip_gauss_hill_integrals": (C.c_int, [vp, C.c_longlong, vp, C.c_int, vp, C.c_double, vp]),
    "edm_hip_gauss_write": (C.c_int, [vp, C.c_char_p]),
    "edm_hip_gauss_multi_write": (C.c_int, [vp, C.c_char_p, C.c_int]),
    "edm_hip_gauss_multi_write_box": (C.c_int, [vp, C.c_char_p, c_dp, c_dp, c_ip, C.c_int]),
    "edm_hip_gauss_add_grid": (C.c_int, [vp, vp, C.c_double, C.c_double]),
    "edm_hip_gauss_add_gauss": (C.c_int, [vp, vp, C.c_double, C.c_double]),
    "edm_hip_gauss_add_from_file": (C.c_int, [vp, C.c_char_p, C.c_double, C.c_double]),
    "edm_hip_bias_create": (C.c_int, [C.POINTER(vp), C.c_char_p]),
    "edm_hip_bias_destroy": (C.c_int, [vp]),
    "edm_hip_bias_setup": (C.c_int, [vp, C.c_double, C.c_double]),
    "edm_hip_bias_subdivide": (C.c_int, [vp, c_dp, c_dp, c_dp, c_dp, c_ip, c_dp]),
    "edm_hip_bias_set_mask": (C.c_int, [vp, vp]),
    "edm_hip_bias_update_forces": (C.c_int, [vp, C.c_longlong, vp, C.c_int, vp, C.c_int, C.c_int, c_dp]),
    "edm_hip_bias_pair_forces": (C.c_int, [vp, C.c_longlong, vp, vp, c_dp]),
    "edm_hip_bias_add_hills": (C.c_int, [vp, C.c_longlong, vp, C.c_int, vp, C.c_int, C.c_longlong]),
    "edm_hip_bias_set_device_rng": (C.c_int, [vp, C.c_int, C.c_ulonglong]),
    "edm_hip_bias_pair_list_upload": (C.c_int, [vp, C.c_longlong, vp, vp, C.c_longlong, vp]),
    "edm_hip_bias_pair_list_step": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, C.c_longlong, c_dp,
                                              C.POINTER(C.c_longlong)]),
    "edm_hip_bias_step": (C.c_int, [vp, C.c_longlong, vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_longlong, c_dp]),
    "edm_hip_bias_pair_step": (C.c_int, [vp, C.c_longlong, vp, vp, C.c_longlong, vp, vp, C.c_longlong, c_dp]),
    "edm_hip_bias_pair_step_host": (C.c_int, [vp, C.c_longlong, vp, vp, C.c_longlong, vp, vp, C.c_longlong, c_dp]),
    "edm_hip_gauss_wait": (C.c_int, [vp]),
    "edm_hip_debug_flag_order_stress": (C.c_int, [C.c_int, C.c_longlong, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "edm_hip_bias_wait": (C.c_int, [vp]),
    "edm_hip_bias_step_host": (C.c_int, [vp, C.c_longlong, vp, C.c_int, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_longlong, c_dp]),
    "edm_hip_bias_pair_step_ordered": (C.c_int, [vp, C.c_longlong, vp, vp, vp, C.c_longlong, vp, vp, C.c_longlong, c_dp]),
    "edm_hip_bias_pair_step_ordered_host": (C.c_int, [vp, C.c_longlong, vp, vp, vp, C.c_longlong, vp, vp, C.c_longlong, c_dp]),
    "edm_hip_bias_pre_add_hill": (C.c_int, [vp, C.c_longlong]),
    "edm_hip_bias_add_hill": (C.c_int, [vp, c_dp, C.c_double]),
    "edm_hip_bias_post_add_hill": (C.c_int, [vp]),
    "edm_hip_bias_write_bias": (C.c_int, [vp, C.c_char_p, C.c_int]),
    "edm_hip_bias_write_lammps_table": (C.c_int, [vp, C.c_char_p, C.c_int]),
    "edm_hip_bias_write_histogram": (C.c_int, [vp, C.c_int]),
    "edm_hip_bias_clear_histogram": (C.c_int, [vp]),
    "edm_hip_bias_gauss": (vp, [vp]),
    "edm_hip_bias_histogram": (vp, [vp]),
    "edm_hip_bias_get": (C.c_int, [vp, C.c_char_p, c_dp]),
    "edm_hip_bias_set": (C.c_int, [vp, C.c_char_p, C.c_double]),
    "edm_hip_bias_get_array": (C.c_int, [vp, C.c_char_p, c_dp]),
    "edm_hip_bias_set_hill_log": (C.c_int, [vp, C.c_int]),
    "edm_hip_comm_unique_id": (C.c_int, [vp, C.c_size_t]),
    "edm_hip_bias_comm_init": (C.c_int, [vp, vp, C.c_int, C.c_int]),
    "edm_hip_bias_comm_init_shm": (C.c_int, [vp, C.c_char_p, C.c_int, C.c_int]),
    "edm_hip_bias_comm_destroy": (C.c_int, [vp]),
}

_dll = None


def exported_symbols():
    """Names the C ABI declares (parsed from include/edm_hip.h)."""
    import re

    text = open(HEADER).read()
    return sorted(set(re.findall(r"\b(edm_hip_[a-z0-9_]+)\s*\(", text)))


def lib():
    """Loads libedm_hip.so (no GPU needed to load; compute calls need one)."""
    global _dll
    if _dll is None:
        if not os.path.exists(LIB_PATH):
            raise EdmHipError("%s is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950)" % LIB_PATH)
        _dll = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            f = getattr(_dll, name)
            f.restype = res
            f.argtypes = args
    return _dll


def check(rc):
    if rc != 0:
        raise EdmHipError("edm_hip status %d: %s" % (rc, lib().edm_hip_last_error().decode(errors="replace")))


def device_count():
    n = C.c_int(0)
    rc = lib().edm_hip_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def require_gpu():
    if device_count() < 1:
        raise EdmHipError("no MI355X visible: the EDM hot path has no CPU fallback")


def device_info():
    buf = C.create_string_buffer(256)
    cu = C.c_int(0)
    mem = C.c_size_t(0)
    check(lib().edm_hip_device_info(buf, 256, C.byref(cu), C.byref(mem)))
    return buf.value.decode(), cu.value, mem.value


def synchronize():
    check(lib().edm_hip_device_synchronize())


class DeviceArray:
    """A typed allocation in HBM."""

    def __init__(self, shape, dtype=np.float64):
        self.shape = tuple(np.atleast_1d(shape).tolist()) if not isinstance(shape, tuple) else shape
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = vp()
        check(lib().edm_hip_malloc(C.byref(p), max(self.nbytes, 8)))
        self.ptr = p.value

    @classmethod
    def from_host(cls, a):
        a = np.ascontiguousarray(a)
        d = cls(a.shape, a.dtype)
        if a.nbytes:
            check(lib().edm_hip_memcpy_h2d(d.ptr, a.ctypes.data, a.nbytes))
        return d

    @classmethod
    def zeros(cls, shape, dtype=np.float64):
        d = cls(shape, dtype)
        if d.nbytes:
            check(lib().edm_hip_memset(d.ptr, 0, d.nbytes))
        return d

    def to_host(self):
        a = np.empty(self.shape, self.dtype)
        if self.nbytes:
            check(lib().edm_hip_memcpy_d2h(a.ctypes.data, self.ptr, self.nbytes))
        return a

    def copy_from(self, a):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        assert a.nbytes == self.nbytes
        check(lib().edm_hip_memcpy_h2d(self.ptr, a.ctypes.data, a.nbytes))

    def free(self):
        if getattr(self, "ptr", None):
            lib().edm_hip_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001
            pass


def _dp(a):
    return a.ctypes.data_as(c_dp)


def _vec(x, n=3):
    a = np.zeros(n)
    x = np.atleast_1d(np.asarray(x, dtype=np.float64))
    a[: x.size] = x
    return a


def _ivec(x, n=3):
    a = np.zeros(n, dtype=np.int32)
    x = np.atleast_1d(np.asarray(x, dtype=np.int32))
    a[: x.size] = x
    return a


def _ptr(d):
    if d is None:
        return None
    return d.ptr if isinstance(d, DeviceArray) else d


class _Geom:
    def _geometry(self, fn):
        g = Geometry()
        check(fn(self.h, C.byref(g)))
        return g

    @property
    def dim(self):
        return self.geometry.dim

    @property
    def size(self):
        return int(self.geometry.total)

    number = property(lambda s: np.array(s.geometry.n[: s.dim], dtype=np.int64))
    dx = property(lambda s: np.array(s.geometry.dx[: s.dim]))
    min = property(lambda s: np.array(s.geometry.min[: s.dim]))
    max = property(lambda s: np.array(s.geometry.max[: s.dim]))
    periodic = property(lambda s: np.array(s.geometry.periodic[: s.dim], dtype=np.int64))


class Grid(_Geom):
    """Device-resident DimmedGrid<DIM> without interpolation (CV histogram)."""

    def __init__(self, handle, owned=True, keep=None):
        self.h, self.owned, self._keep = handle, owned, keep

    @classmethod
    def create(cls, lo, hi, spacing, periodic, b_derivatives=0, b_interpolate=0):
        """make_grid (grid.h:911)"""
        dim = len(np.atleast_1d(lo))
        h = vp()
        check(lib().edm_hip_grid_create_ex(C.byref(h), dim, _dp(_vec(lo)), _dp(_vec(hi)), _dp(_vec(spacing)),
                                           _ivec(periodic).ctypes.data_as(c_ip), b_derivatives, b_interpolate))
        return cls(h.value)

    @classmethod
    def read_file(cls, dim, filename, b_interpolate=1):
        """read_grid (grid.h:923, :928)"""
        h = vp()
        check(lib().edm_hip_grid_read(C.byref(h), dim, os.fsencode(filename), b_interpolate))
        return cls(h.value)

    def read(self, filename):
        """Grid::read (grid.h:712-835)"""
        check(lib().edm_hip_grid_reread(self.h, os.fsencode(filename)))

    def set_interpolation(self, b):
        check(lib().edm_hip_grid_set_interpolation(self.h, int(b)))

    has_derivatives = property(lambda s: bool(s.geometry.derivatives))

    @property
    def derivs(self):
        d = np.empty((self.size, self.dim))
        check(lib().edm_hip_grid_download_derivs(self.h, _dp(d)))
        return d

    def upload_derivs(self, values, derivs):
        v = np.ascontiguousarray(values, dtype=np.float64)
        d = np.ascontiguousarray(derivs, dtype=np.float64)
        check(lib().edm_hip_grid_upload_derivs(self.h, _dp(v), _dp(d)))

    def get_value_deriv(self, x):
        """DimmedGrid::get_value_deriv batched: x [n, >=dim] -> (V[n], der[n, dim])"""
        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64)
        n, stride = x.shape
        dx = DeviceArray.from_host(x)
        dE = DeviceArray((n,))
        dD = DeviceArray((n, self.dim))
        check(lib().edm_hip_grid_get_value_deriv(self.h, n, dx.ptr, stride, dE.ptr, dD.ptr))
        return dE.to_host(), dD.to_host()

    def add(self, other, scale=1.0, offset=0.0):
        """Grid::add (grid.h:275-290)"""
        fn = lib().edm_hip_grid_add_gauss if isinstance(other, Gauss) else lib().edm_hip_grid_add_grid
        check(fn(self.h, other.h, scale, offset))

    def __del__(self):
        try:
            if getattr(self, "owned", False) and self.h:
                lib().edm_hip_grid_destroy(self.h)
                self.h = None
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    @property
    def geometry(self):
        return self._geometry(lib().edm_hip_grid_geometry)

    @property
    def values(self):
        a = np.empty(self.size)
        check(lib().edm_hip_grid_download(self.h, _dp(a)))
        return a

    def upload(self, values):
        a = np.ascontiguousarray(values, dtype=np.float64)
        check(lib().edm_hip_grid_upload(self.h, _dp(a)))

    def clear(self):
        check(lib().edm_hip_grid_clear(self.h))

    def add_values(self, d_x, n, stride, d_w=None, w_const=1.0):
        check(lib().edm_hip_grid_add_values(self.h, n, _ptr(d_x), stride, _ptr(d_w), w_const))

    def write(self, filename):
        check(lib().edm_hip_grid_write(self.h, os.fsencode(filename)))

    def multi_write(self, filename, box_min, box_max, periodic, lammps=0):
        check(lib().edm_hip_grid_multi_write(self.h, os.fsencode(filename), _dp(_vec(box_min)), _dp(_vec(box_max)),
                                             _ivec(periodic).ctypes.data_as(c_ip), lammps))


class Gauss(_Geom):
    """Device-resident DimmedGaussGrid<DIM>."""

    def __init__(self, handle, owned=True, keep=None):
        self.h, self.owned, self._keep = handle, owned, keep

    @classmethod
    def create(cls, lo, hi, spacing, periodic, b_interp, sigma):
        dim = len(np.atleast_1d(lo))
        h = vp()
        check(lib().edm_hip_gauss_create(C.byref(h), dim, _dp(_vec(lo)), _dp(_vec(hi)), _dp(_vec(spacing)),
                                         _ivec(periodic).ctypes.data_as(c_ip), b_interp, _dp(_vec(sigma))))
        return cls(h.value)

    @classmethod
    def read_file(cls, dim, filename, sigma):
        """read_gauss_grid (gaussian_grid.h:647)"""
        h = vp()
        check(lib().edm_hip_gauss_read(C.byref(h), dim, os.fsencode(filename), _dp(_vec(sigma))))
        return cls(h.value)

    def read(self, filename):
        """GaussGrid::read (gaussian_grid.h:140-142)"""
        check(lib().edm_hip_gauss_reread(self.h, os.fsencode(filename)))

    def set_interpolation(self, b):
        check(lib().edm_hip_gauss_set_interpolation(self.h, int(b)))

    def set_lookup_replica(self, mode):
        """-1 automatic, 0 off, 1 always (edm_hip_gauss_set_lookup_replica)"""
        check(lib().edm_hip_gauss_set_lookup_replica(self.h, int(mode)))

    def lookup_replica_info(self):
        """(in use, bytes, full rebuilds so far)"""
        u, b, r = C.c_int(0), C.c_longlong(0), C.c_longlong(0)
        check(lib().edm_hip_gauss_lookup_replica_info(self.h, C.byref(u), C.byref(b), C.byref(r)))
        return bool(u.value), b.value, r.value

    def add(self, other, scale=1.0, offset=0.0):
        """Grid::add (grid.h:275-290) with another device grid"""
        fn = lib().edm_hip_gauss_add_gauss if isinstance(other, Gauss) else lib().edm_hip_gauss_add_grid
        check(fn(self.h, other.h, scale, offset))

    def multi_write_box(self, filename, box_min, box_max, periodic, lammps=0):
        check(lib().edm_hip_gauss_multi_write_box(self.h, os.fsencode(filename), _dp(_vec(box_min)), _dp(_vec(box_max)),
                                                  _ivec(periodic).ctypes.data_as(c_ip), lammps))

    def __del__(self):
        try:
            if getattr(self, "owned", False) and self.h:
                lib().edm_hip_gauss_destroy(self.h)
                self.h = None
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    @property
    def geometry(self):
        return self._geometry(lib().edm_hip_gauss_geometry)

    sigma = property(lambda s: np.array(s.geometry.sigma[: s.dim]))
    minisize = property(lambda s: [int(v) for v in s.geometry.minisize[: s.dim]])
    boundary_min = property(lambda s: np.array(s.geometry.boundary_min[: s.dim]))
    boundary_max = property(lambda s: np.array(s.geometry.boundary_max[: s.dim]))
    boundary_periodic = property(lambda s: np.array(s.geometry.boundary_periodic[: s.dim], dtype=np.int64))

    def set_boundary(self, lo, hi, periodic):
        check(lib().edm_hip_gauss_set_boundary(self.h, _dp(_vec(lo)), _dp(_vec(hi)), _ivec(periodic).ctypes.data_as(c_ip)))

    def bc_tables(self, dim_index):
        """the two McGovern-De Pablo tables of one non-periodic boundary dimension, read back from HBM"""
        t0, t1 = np.empty(65536), np.empty(65536)
        check(lib().edm_hip_gauss_download_tables(self.h, dim_index, _dp(t0), _dp(t1)))
        return t0, t1

    def download(self):
        v = np.empty(self.size)
        d = np.empty((self.size, self.dim))
        check(lib().edm_hip_gauss_download(self.h, _dp(v), _dp(d)))
        return v, d

    def upload(self, values, derivs):
        v = np.ascontiguousarray(values, dtype=np.float64)
        d = np.ascontiguousarray(derivs, dtype=np.float64)
        check(lib().edm_hip_gauss_upload(self.h, _dp(v), _dp(d)))

    def clear(self):
        check(lib().edm_hip_gauss_clear(self.h))

    def device_buffer(self):
        p = vp()
        r = C.c_int(0)
        n = C.c_longlong(0)
        check(lib().edm_hip_gauss_device_buffer(self.h, C.byref(p), C.byref(r), C.byref(n)))
        return p.value, r.value, n.value

    # ---- batched lookups (positions: host [n, stride] float64, uploaded here) ----
    def get_value_deriv(self, x):
        """x: [n, >=dim] -> (E[n], der[n, dim])"""
        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64)
        n, stride = x.shape
        dx = DeviceArray.from_host(x)
        dE = DeviceArray((n,))
        dD = DeviceArray((n, self.dim))
        check(lib().edm_hip_gauss_get_value_deriv(self.h, n, dx.ptr, stride, dE.ptr, dD.ptr))
        return dE.to_host(), dD.to_host()

    def sample_index(self, x):
        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64)
        n, stride = x.shape
        dx = DeviceArray.from_host(x)
        df = DeviceArray((n,), np.int64)
        check(lib().edm_hip_gauss_sample_index(self.h, n, dx.ptr, stride, df.ptr))
        return df.to_host()

    def remap(self, x):
        """DimmedGaussGrid::remap batched: x [n, >=dim] -> remapped [n, dim]"""
        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64)
        n, stride = x.shape
        dx = DeviceArray.from_host(x)
        do = DeviceArray((n, self.dim))
        check(lib().edm_hip_gauss_remap(self.h, n, dx.ptr, stride, do.ptr))
        return do.to_host()

    def update_forces(self, x, f, mask=None, apply_mask=-1):
        """EDMBias::update_forces on host arrays (f updated in place); returns the energy."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        n, stride = x.shape
        dx = DeviceArray.from_host(x)
        df = DeviceArray.from_host(f)
        dm = DeviceArray.from_host(np.ascontiguousarray(mask, dtype=np.int32)) if mask is not None else None
        e = C.c_double(0)
        check(lib().edm_hip_gauss_update_forces(self.h, n, dx.ptr, stride, df.ptr, f.shape[1], _ptr(dm), apply_mask, C.byref(e)))
        f[...] = df.to_host()
        return e.value

    def pair_forces(self, r):
        r = np.ascontiguousarray(r, dtype=np.float64)
        dr = DeviceArray.from_host(r)
        df = DeviceArray((r.size,))
        e = C.c_double(0)
        check(lib().edm_hip_gauss_pair_forces(self.h, r.size, dr.ptr, df.ptr, C.byref(e)))
        return e.value, df.to_host()

    def pair_forces_device(self, d_r, d_f, n):
        e = C.c_double(0)
        check(lib().edm_hip_gauss_pair_forces(self.h, n, _ptr(d_r), _ptr(d_f), C.byref(e)))
        return e.value

    def profile_enable(self, enabled=True):
        check(lib().edm_hip_gauss_profile_enable(self.h, int(enabled)))

    def profile_read(self, reset=True):
        """(total kernel ms, launches) of the lookup kernel, from HIP events on the handle's stream."""
        ms = C.c_double(0)
        n = C.c_longlong(0)
        check(lib().edm_hip_gauss_profile_read(self.h, C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value

    # ---- hills ----
    def add_values(self, x, h):
        """Batched GaussGrid::add_value in list order; returns per-hill bias_added."""
        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64)
        n, stride = x.shape
        hh = np.ascontiguousarray(np.broadcast_to(np.asarray(h, dtype=np.float64), (n,)))
        dx = DeviceArray.from_host(x)
        dh = DeviceArray.from_host(hh)
        da = DeviceArray((n,))
        tot = C.c_double(0)
        check(lib().edm_hip_gauss_add_values(self.h, n, dx.ptr, stride, dh.ptr, 0.0, da.ptr, C.byref(tot)))
        return da.to_host()

    def add_value(self, x, h):
        return float(self.add_values(np.atleast_2d(np.asarray(x, dtype=np.float64)), [h])[0])

    def hill_integrals(self, x, h):
        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64)
        n, stride = x.shape
        hh = np.ascontiguousarray(np.broadcast_to(np.asarray(h, dtype=np.float64), (n,)))
        dx = DeviceArray.from_host(x)
        dh = DeviceArray.from_host(hh)
        da = DeviceArray((n,))
        check(lib().edm_hip_gauss_hill_integrals(self.h, n, dx.ptr, stride, dh.ptr, 0.0, da.ptr))
        return da.to_host()

    def write(self, filename):
        check(lib().edm_hip_gauss_write(self.h, os.fsencode(filename)))

    def multi_write(self, filename, lammps=0):
        check(lib().edm_hip_gauss_multi_write(self.h, os.fsencode(filename), lammps))

    def add_from_file(self, filename, scale=1.0, offset=0.0):
        check(lib().edm_hip_gauss_add_from_file(self.h, os.fsencode(filename), scale, offset))


class Bias:
    """EDMBias over the C ABI."""

    def __init__(self, config_path):
        h = vp()
        rc = lib().edm_hip_bias_create(C.byref(h), os.fsencode(config_path))
        self.h = h.value
        check(rc)
        self._mask = None

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().edm_hip_bias_destroy(self.h)
                self.h = None
        except Exception:  # noqa: BLE001  (interpreter shutdown: module globals may already be gone)
            pass

    def setup(self, temperature, boltzmann):
        check(lib().edm_hip_bias_setup(self.h, temperature, boltzmann))

    def subdivide(self, sublo, subhi, boxlo, boxhi, periodic, skin):
        check(lib().edm_hip_bias_subdivide(self.h, _dp(_vec(sublo)), _dp(_vec(subhi)), _dp(_vec(boxlo)), _dp(_vec(boxhi)),
                                           _ivec(periodic).ctypes.data_as(c_ip), _dp(_vec(skin))))

    def set_mask(self, mask):
        self._mask = DeviceArray.from_host(np.ascontiguousarray(mask, dtype=np.int32))
        check(lib().edm_hip_bias_set_mask(self.h, self._mask.ptr))

    def update_forces(self, x, f, apply_mask=-1):
        x = np.ascontiguousarray(x, dtype=np.float64)
        dx = DeviceArray.from_host(x)
        df = DeviceArray.from_host(f)
        e = C.c_double(0)
        check(lib().edm_hip_bias_update_forces(self.h, x.shape[0], dx.ptr, x.shape[1], df.ptr, f.shape[1], apply_mask, C.byref(e)))
        f[...] = df.to_host()
        return e.value

    def update_forces_device(self, d_x, n, x_stride, d_f, f_stride, apply_mask=-1):
        e = C.c_double(0)
        check(lib().edm_hip_bias_update_forces(self.h, n, _ptr(d_x), x_stride, _ptr(d_f), f_stride, apply_mask, C.byref(e)))
        return e.value

    def pair_forces_device(self, d_r, d_f, n):
        e = C.c_double(0)
        check(lib().edm_hip_bias_pair_forces(self.h, n, _ptr(d_r), _ptr(d_f), C.byref(e)))
        return e.value

    def set_device_rng(self, enabled, seed):
        """acceptance uniforms drawn on the device (fast mode): pass d_u = None afterwards"""
        check(lib().edm_hip_bias_set_device_rng(self.h, int(enabled), int(seed)))

    def pair_list_upload(self, pair_i, pair_j, types):
        """flattened half list (host int32 arrays, neighbour-list order) + atom types -> device-resident list"""
        pi = np.ascontiguousarray(pair_i, dtype=np.int32)
        pj = np.ascontiguousarray(pair_j, dtype=np.int32)
        ty = np.ascontiguousarray(types, dtype=np.int32)
        check(lib().edm_hip_bias_pair_list_upload(self.h, len(pi), pi.ctypes.data, pj.ctypes.data, len(ty), ty.ctypes.data))

    def pair_list_step_device(self, nlocal, itype, jtype, d_x, d_fdelta, hill_step, est):
        """fix edm_pair over the uploaded list: returns (energy, add_hill calls of the step)"""
        e = C.c_double(0)
        nc = C.c_longlong(0)
        check(lib().edm_hip_bias_pair_list_step(self.h, nlocal, itype, jtype, _ptr(d_x), _ptr(d_fdelta), int(hill_step), est,
                                                C.byref(e), C.byref(nc)))
        return e.value, nc.value

    def step_device(self, d_x, x_stride, d_f, f_stride, n, d_u, apply_mask=-1, est=-1):
        """update_forces + add_hills over the same samples (one hill-depositing fix edm step), one host wait"""
        e = C.c_double(0)
        check(lib().edm_hip_bias_step(self.h, n, _ptr(d_x), x_stride, _ptr(d_f), f_stride, _ptr(d_u), apply_mask, est,
                                      C.byref(e)))
        return e.value

    def pair_step_device(self, d_r, d_f, n, d_sample_r, d_u, n_samples, est=-1):
        """update_pair_forces + add_pair_hills of one hill-depositing fix edm_pair step, one host wait"""
        e = C.c_double(0)
        check(lib().edm_hip_bias_pair_step(self.h, n, _ptr(d_r), _ptr(d_f), n_samples, _ptr(d_sample_r), _ptr(d_u), est,
                                           C.byref(e)))
        return e.value

    def add_hills(self, x, runiform, apply_mask=-1, est=-1):
        x = np.ascontiguousarray(x, dtype=np.float64)
        dx = DeviceArray.from_host(x)
        du = DeviceArray.from_host(np.ascontiguousarray(runiform, dtype=np.float64))
        check(lib().edm_hip_bias_add_hills(self.h, x.shape[0], dx.ptr, x.shape[1], du.ptr, apply_mask, est))

    def add_hills_device(self, d_x, n, x_stride, d_u, apply_mask=-1, est=-1):
        check(lib().edm_hip_bias_add_hills(self.h, n, _ptr(d_x), x_stride, _ptr(d_u), apply_mask, est))

    def pre_add_hill(self, est):
        check(lib().edm_hip_bias_pre_add_hill(self.h, est))

    def add_hill(self, position, runiform):
        check(lib().edm_hip_bias_add_hill(self.h, _dp(_vec(position)), runiform))

    def post_add_hill(self):
        check(lib().edm_hip_bias_post_add_hill(self.h))

    def write_bias(self, filename, serial_format=1):
        check(lib().edm_hip_bias_write_bias(self.h, os.fsencode(filename), serial_format))

    def write_lammps_table(self, filename, serial_format=1):
        check(lib().edm_hip_bias_write_lammps_table(self.h, os.fsencode(filename), serial_format))

    def write_histogram(self, serial_format=1):
        check(lib().edm_hip_bias_write_histogram(self.h, serial_format))

    def clear_histogram(self):
        check(lib().edm_hip_bias_clear_histogram(self.h))

    @property
    def gauss(self):
        return Gauss(lib().edm_hip_bias_gauss(self.h), owned=False, keep=self)

    @property
    def hist(self):
        return Grid(lib().edm_hip_bias_histogram(self.h), owned=False, keep=self)

    def get(self, name):
        v = C.c_double(0)
        check(lib().edm_hip_bias_get(self.h, name.encode(), C.byref(v)))
        return v.value

    def set(self, name, value):
        check(lib().edm_hip_bias_set(self.h, name.encode(), float(value)))

    def array(self, name):
        out = np.zeros(3)
        check(lib().edm_hip_bias_get_array(self.h, name.encode(), _dp(out)))
        return out[: int(self.get("dim"))].copy()

    def set_hill_log(self, enabled):
        check(lib().edm_hip_bias_set_hill_log(self.h, int(enabled)))

    def comm_init(self, id_bytes, nranks, rank):
        buf = C.create_string_buffer(bytes(id_bytes), 128) if id_bytes is not None else None
        check(lib().edm_hip_bias_comm_init(self.h, C.cast(buf, vp) if buf is not None else None, nranks, rank))


def pinned_array(n, dtype=np.float64):
    """numpy array over page-locked host memory (edm_hip_host_malloc); keep the returned array alive"""
    dt = np.dtype(dtype)
    p = vp()
    check(lib().edm_hip_host_malloc(C.byref(p), max(1, n) * dt.itemsize))
    buf = (C.c_char * (max(1, n) * dt.itemsize)).from_address(p.value)
    a = np.frombuffer(buf, dtype=dt, count=n)
    _pinned_keep.append((p.value, buf))
    return a


_pinned_keep = []


def _bias_pair_step_host(self, r, force, sample_r, runiform, est=-1):
    """edm_hip_bias_pair_step_host on host numpy arrays (force is written in place); returns the energy"""
    e = C.c_double(0)
    ns = 0 if sample_r is None else len(sample_r)
    check(lib().edm_hip_bias_pair_step_host(self.h, len(r), r.ctypes.data, force.ctypes.data, ns,
                                            None if sample_r is None else sample_r.ctypes.data,
                                            None if runiform is None else runiform.ctypes.data, est, C.byref(e)))
    return e.value


Bias.pair_step_host = _bias_pair_step_host


def _bias_pair_step_ordered_device(self, d_r, d_f, d_first_sample, n, d_sample_r, d_u, n_samples, est=-1):
    """one hill step of fix edm_pair in the reference's own order (edm_hip_bias_pair_step_ordered); returns the energy"""
    e = C.c_double(0)
    check(lib().edm_hip_bias_pair_step_ordered(self.h, n, _ptr(d_r), _ptr(d_f), _ptr(d_first_sample), n_samples,
                                               _ptr(d_sample_r), _ptr(d_u), est, C.byref(e)))
    return e.value


def _bias_pair_step_ordered_host(self, r, force, first_sample, sample_r, runiform, est=-1):
    """edm_hip_bias_pair_step_ordered_host on host numpy arrays (force written in place); returns the energy"""
    e = C.c_double(0)
    ns = 0 if sample_r is None else len(sample_r)
    fs = np.ascontiguousarray(first_sample, dtype=np.int32)
    check(lib().edm_hip_bias_pair_step_ordered_host(self.h, len(r), r.ctypes.data, force.ctypes.data, fs.ctypes.data, ns,
                                                    None if sample_r is None else sample_r.ctypes.data,
                                                    None if runiform is None else runiform.ctypes.data, est, C.byref(e)))
    return e.value


def _bias_step_host(self, x, f, mask=None, runiform=None, apply_mask=-1, hill_step=True, est=-1):
    """edm_hip_bias_step_host on host numpy arrays x [n, xs], f [n, fs] (f updated in place); returns the energy"""
    e = C.c_double(0)
    assert x.flags.c_contiguous and f.flags.c_contiguous and x.dtype == np.float64 and f.dtype == np.float64
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.int32)
    check(lib().edm_hip_bias_step_host(self.h, x.shape[0], x.ctypes.data, x.shape[1], f.ctypes.data, f.shape[1],
                                       None if m is None else m.ctypes.data,
                                       None if runiform is None else runiform.ctypes.data, apply_mask, int(hill_step), est,
                                       C.byref(e)))
    return e.value


Bias.step_host = _bias_step_host
Bias.wait = lambda self: check(lib().edm_hip_bias_wait(self.h))
Gauss.wait = lambda self: check(lib().edm_hip_gauss_wait(self.h))


def flag_order_stress(iterations, words):
    """edm_hip_debug_flag_order_stress: (words found older than their completion flag, launches whose flag timed out)"""
    bad, late = C.c_longlong(0), C.c_longlong(0)
    check(lib().edm_hip_debug_flag_order_stress(int(iterations), int(words), C.byref(bad), C.byref(late)))
    return bad.value, late.value
Bias.pair_step_ordered_device = _bias_pair_step_ordered_device
Bias.pair_step_ordered_host = _bias_pair_step_ordered_host


def _bias_comm_init_shm(self, shm_name, nranks, rank):
    """the exchange staged through POSIX shared memory (edm_hip_bias_comm_init_shm)"""
    check(lib().edm_hip_bias_comm_init_shm(self.h, shm_name.encode(), nranks, rank))


Bias.comm_init_shm = _bias_comm_init_shm


def comm_unique_id():
    buf = C.create_string_buffer(128)
    check(lib().edm_hip_comm_unique_id(C.cast(buf, vp), 128))
    return buf.raw
