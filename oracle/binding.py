"""ctypes binding for the test oracle (oracle/liboracle.so, prefix ``ora_``) and,
where it has been built, the real reference (oracle/_ref/libedm_ref.so, prefix
``ref_``).  Both export the same C API (oracle/edm_oracle.h).

TEST INFRASTRUCTURE ONLY: imported by tests/, oracle/gen_golden.py,
__graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the
product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libedm_ref.so")
REF_MPI_SO = os.path.join(HERE, "_ref", "libedm_ref_mpi.so")   # the reference WITHOUT -DEDM_SERIAL (make ref_mpi)

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)
c_zp = C.POINTER(C.c_size_t)


def build_oracle(force=False):
    """Compile the C restatement (and the reference build when its sources exist)."""
    if force or not os.path.exists(ORACLE_SO) or (
        os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(HERE, "edm_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/lib") and (force or not os.path.exists(REF_SO)):
        subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)


def _dp(a):
    return a.ctypes.data_as(c_dp)


def _vec(x, n=None):
    a = np.ascontiguousarray(np.atleast_1d(np.asarray(x, dtype=np.float64)))
    if n is not None and a.size < n:
        a = np.concatenate([a, np.zeros(n - a.size)])
    return a


def _ivec(x):
    return np.ascontiguousarray(np.atleast_1d(np.asarray(x, dtype=np.int32)))


class Lib:
    """One loaded library + prefix."""

    def __init__(self, path, prefix):
        self.path = path
        self.prefix = prefix
        self.dll = C.CDLL(path)
        self._proto()

    def fn(self, name):
        return getattr(self.dll, self.prefix + name)

    def _set(self, name, restype, argtypes):
        f = self.fn(name)
        f.restype = restype
        f.argtypes = argtypes

    def _proto(self):
        vp = C.c_void_p
        s = self._set
        s("grid_create", vp, [C.c_int, c_dp, c_dp, c_dp, c_ip, C.c_int, C.c_int])
        s("grid_read", vp, [C.c_int, C.c_char_p, C.c_int])
        s("grid_free", None, [vp])
        s("grid_dim", C.c_int, [vp])
        s("grid_size", C.c_size_t, [vp])
        s("grid_number", c_ip, [vp])
        s("grid_dx", c_dp, [vp])
        s("grid_min", c_dp, [vp])
        s("grid_max", c_dp, [vp])
        s("grid_periodic", c_ip, [vp])
        s("grid_has_deriv", C.c_int, [vp])
        s("grid_values", c_dp, [vp])
        s("grid_derivs", c_dp, [vp])
        s("grid_set_interpolation", None, [vp, C.c_int])
        s("grid_get_index", None, [vp, c_dp, c_zp])
        s("grid_multi2one", C.c_size_t, [vp, c_zp])
        s("grid_one2multi", None, [vp, C.c_size_t, c_zp])
        s("grid_in_grid", C.c_int, [vp, c_dp])
        s("grid_get_value", C.c_double, [vp, c_dp])
        s("grid_get_value_deriv", C.c_double, [vp, c_dp, c_dp])
        s("grid_add_value", C.c_double, [vp, c_dp, C.c_double])
        s("grid_clear", None, [vp])
        s("grid_max_value", C.c_double, [vp])
        s("grid_min_value", C.c_double, [vp])
        s("grid_expected_bias", C.c_double, [vp])
        s("grid_add_grid", None, [vp, vp, C.c_double, C.c_double])
        s("grid_write", None, [vp, C.c_char_p])
        s("grid_multi_write", None, [vp, C.c_char_p, c_dp, c_dp, c_ip, C.c_int])
        s("gauss_create", vp, [C.c_int, c_dp, c_dp, c_dp, c_ip, C.c_int, c_dp])
        s("gauss_read", vp, [C.c_int, C.c_char_p, c_dp])
        s("gauss_free", None, [vp])
        s("gauss_grid", vp, [vp])
        s("gauss_set_boundary", None, [vp, c_dp, c_dp, c_ip])
        s("gauss_add_value", C.c_double, [vp, c_dp, C.c_double])
        s("gauss_add_values", C.c_double, [vp, C.c_longlong, c_dp, C.c_int, C.c_double])
        s("gauss_get_value", C.c_double, [vp, c_dp])
        s("gauss_get_value_deriv", C.c_double, [vp, c_dp, c_dp])
        s("gauss_remap", None, [vp, c_dp])
        s("gauss_in_bounds", C.c_int, [vp, c_dp])
        s("gauss_get_volume", C.c_double, [vp])
        s("gauss_sigma", c_dp, [vp])
        s("gauss_minisize", c_zp, [vp])
        s("gauss_minisize_total", C.c_size_t, [vp])
        s("gauss_bc_table", c_dp, [vp, C.c_int, C.c_int])
        s("gauss_boundary_min", c_dp, [vp])
        s("gauss_boundary_max", c_dp, [vp])
        s("gauss_boundary_periodic", c_ip, [vp])
        s("gauss_write", None, [vp, C.c_char_p])
        s("gauss_multi_write", None, [vp, C.c_char_p, C.c_int])
        s("bias_create", vp, [C.c_char_p])
        s("bias_free", None, [vp])
        s("bias_setup", None, [vp, C.c_double, C.c_double])
        s("bias_subdivide", None, [vp, c_dp, c_dp, c_dp, c_dp, c_ip, c_dp])
        s("bias_update_forces", C.c_double, [vp, C.c_int, c_dp, c_dp, C.c_int, C.c_int])
        s("bias_update_force", C.c_double, [vp, c_dp, c_dp])
        s("bias_set_mask", None, [vp, c_ip])
        s("bias_add_hills", None, [vp, C.c_int, c_dp, C.c_int, c_dp, C.c_int])
        s("bias_pre_add_hill", None, [vp, C.c_int])
        s("bias_add_hill", None, [vp, c_dp, C.c_double])
        s("bias_post_add_hill", None, [vp])
        s("bias_pair_loop", C.c_double, [vp, C.c_int, c_dp, c_ip, c_dp, C.c_int, C.c_int, c_dp, c_ip])
        s("bias_write_bias", None, [vp, C.c_char_p])
        s("bias_write_lammps_table", None, [vp, C.c_char_p])
        s("bias_write_histogram", None, [vp])
        s("bias_clear_histogram", None, [vp])
        s("bias_gauss", vp, [vp])
        s("bias_hist", vp, [vp])
        s("bias_get", C.c_double, [vp, C.c_char_p])
        s("bias_set", None, [vp, C.c_char_p, C.c_double])
        s("bias_array", c_dp, [vp, C.c_char_p])
        if hasattr(self.dll, self.prefix + "mpi_rank"):   # (the reference builds only)
            s("mpi_rank", C.c_int, [])
            s("mpi_size", C.c_int, [])
            s("mpi_barrier", None, [])
            s("mpi_finalize", None, [])
            s("is_mpi_build", C.c_int, [])


_LIBS = {}


def load(kind="oracle"):
    """kind: 'oracle' (C restatement) or 'ref' (the real reference build)."""
    if kind not in _LIBS:
        if kind == "oracle":
            build_oracle()
            _LIBS[kind] = Lib(ORACLE_SO, "ora_")
        elif kind == "ref":
            if not os.path.exists(REF_SO):
                build_oracle()
            if not os.path.exists(REF_SO):
                raise FileNotFoundError(REF_SO)
            _LIBS[kind] = Lib(REF_SO, "ref_")
        elif kind == "ref_mpi":
            if not os.path.exists(REF_MPI_SO) and os.path.isdir("/root/reference/lib"):
                subprocess.check_call(["make", "-C", HERE, "ref_mpi"], stdout=subprocess.DEVNULL)
            if not os.path.exists(REF_MPI_SO):
                raise FileNotFoundError(REF_MPI_SO)
            _LIBS[kind] = Lib(REF_MPI_SO, "ref_")
        else:
            raise ValueError(kind)
    return _LIBS[kind]


def have_ref():
    return os.path.exists(REF_SO) or os.path.isdir("/root/reference/lib")


class Grid:
    def __init__(self, lib, handle, owned=True, keep=None):
        self.lib, self.h, self.owned, self._keep = lib, handle, owned, keep

    @classmethod
    def create(cls, lib, lo, hi, spacing, periodic, b_deriv, b_interp):
        lo, hi, sp, per = _vec(lo), _vec(hi), _vec(spacing), _ivec(periodic)
        h = lib.fn("grid_create")(len(lo), _dp(lo), _dp(hi), _dp(sp), per.ctypes.data_as(c_ip), b_deriv, b_interp)
        return cls(lib, h)

    @classmethod
    def read(cls, lib, dim, filename, b_interp=1):
        return cls(lib, lib.fn("grid_read")(dim, os.fsencode(filename), b_interp))

    def __del__(self):
        if getattr(self, "owned", False) and self.h:
            self.lib.fn("grid_free")(self.h)
            self.h = None

    @property
    def dim(self):
        return self.lib.fn("grid_dim")(self.h)

    @property
    def size(self):
        return self.lib.fn("grid_size")(self.h)

    def _arr(self, name, ctype_n=None):
        p = self.lib.fn(name)(self.h)
        return np.array([p[i] for i in range(self.dim)])

    number = property(lambda s: s._arr("grid_number").astype(np.int64))
    dx = property(lambda s: s._arr("grid_dx"))
    min = property(lambda s: s._arr("grid_min"))
    max = property(lambda s: s._arr("grid_max"))
    periodic = property(lambda s: s._arr("grid_periodic").astype(np.int64))
    has_deriv = property(lambda s: s.lib.fn("grid_has_deriv")(s.h))

    @property
    def values(self):
        """numpy VIEW of the value array (writes go through)."""
        p = self.lib.fn("grid_values")(self.h)
        return np.ctypeslib.as_array(p, shape=(self.size,))

    @property
    def derivs(self):
        p = self.lib.fn("grid_derivs")(self.h)
        return np.ctypeslib.as_array(p, shape=(self.size, self.dim))

    def set_interpolation(self, b):
        self.lib.fn("grid_set_interpolation")(self.h, b)

    def get_index(self, x):
        x = _vec(x, 3)
        out = (C.c_size_t * 3)()
        self.lib.fn("grid_get_index")(self.h, _dp(x), out)
        return [int(out[i]) for i in range(self.dim)]

    def multi2one(self, idx):
        a = (C.c_size_t * 3)(*list(idx) + [0] * (3 - len(idx)))
        return int(self.lib.fn("grid_multi2one")(self.h, a))

    def one2multi(self, index):
        out = (C.c_size_t * 3)()
        self.lib.fn("grid_one2multi")(self.h, index, out)
        return [int(out[i]) for i in range(self.dim)]

    def in_grid(self, x):
        return self.lib.fn("grid_in_grid")(self.h, _dp(_vec(x, 3)))

    def get_value(self, x):
        return self.lib.fn("grid_get_value")(self.h, _dp(_vec(x, 3)))

    def get_value_deriv(self, x):
        der = np.zeros(3)
        v = self.lib.fn("grid_get_value_deriv")(self.h, _dp(_vec(x, 3)), _dp(der))
        return v, der[: self.dim].copy()

    def add_value(self, x, value):
        return self.lib.fn("grid_add_value")(self.h, _dp(_vec(x, 3)), value)

    def clear(self):
        self.lib.fn("grid_clear")(self.h)

    def max_value(self):
        return self.lib.fn("grid_max_value")(self.h)

    def min_value(self):
        return self.lib.fn("grid_min_value")(self.h)

    def expected_bias(self):
        return self.lib.fn("grid_expected_bias")(self.h)

    def add_grid(self, other, scale, offset):
        self.lib.fn("grid_add_grid")(self.h, other.h, scale, offset)

    def write(self, filename):
        self.lib.fn("grid_write")(self.h, os.fsencode(filename))

    def multi_write(self, filename, box_min, box_max, periodic, lammps):
        a, b, p = _vec(box_min, 3), _vec(box_max, 3), _ivec(list(np.atleast_1d(periodic)) + [0, 0, 0])
        self.lib.fn("grid_multi_write")(self.h, os.fsencode(filename), _dp(a), _dp(b), p.ctypes.data_as(c_ip), lammps)


class Gauss:
    def __init__(self, lib, handle, owned=True, keep=None):
        self.lib, self.h, self.owned, self._keep = lib, handle, owned, keep

    @classmethod
    def create(cls, lib, lo, hi, spacing, periodic, b_interp, sigma):
        lo, hi, sp, per, sg = _vec(lo), _vec(hi), _vec(spacing), _ivec(periodic), _vec(sigma)
        h = lib.fn("gauss_create")(len(lo), _dp(lo), _dp(hi), _dp(sp), per.ctypes.data_as(c_ip), b_interp, _dp(sg))
        return cls(lib, h)

    @classmethod
    def read(cls, lib, dim, filename, sigma):
        sg = _vec(sigma, 3)
        return cls(lib, lib.fn("gauss_read")(dim, os.fsencode(filename), _dp(sg)))

    def __del__(self):
        if getattr(self, "owned", False) and self.h:
            self.lib.fn("gauss_free")(self.h)
            self.h = None

    @property
    def grid(self):
        return Grid(self.lib, self.lib.fn("gauss_grid")(self.h), owned=False, keep=self)

    @property
    def dim(self):
        return self.grid.dim

    def set_boundary(self, lo, hi, periodic):
        lo, hi, per = _vec(lo, 3), _vec(hi, 3), _ivec(list(np.atleast_1d(periodic)) + [0, 0, 0])
        self.lib.fn("gauss_set_boundary")(self.h, _dp(lo), _dp(hi), per.ctypes.data_as(c_ip))

    def add_value(self, x, height):
        return self.lib.fn("gauss_add_value")(self.h, _dp(_vec(x, 3)), height)

    def add_values(self, x, height):
        """n add_value calls in one C call; x: float64 [n, stride] C-contiguous.  Returns the summed bias_added."""
        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64)
        return self.lib.fn("gauss_add_values")(self.h, x.shape[0], _dp(x), x.shape[1], height)

    def get_value(self, x):
        return self.lib.fn("gauss_get_value")(self.h, _dp(_vec(x, 3)))

    def get_value_deriv(self, x):
        der = np.zeros(3)
        v = self.lib.fn("gauss_get_value_deriv")(self.h, _dp(_vec(x, 3)), _dp(der))
        return v, der[: self.dim].copy()

    def remap(self, x):
        a = _vec(x, 3).copy()
        self.lib.fn("gauss_remap")(self.h, _dp(a))
        return a[: self.dim].copy()

    def in_bounds(self, x):
        return self.lib.fn("gauss_in_bounds")(self.h, _dp(_vec(x, 3)))

    def get_volume(self):
        return self.lib.fn("gauss_get_volume")(self.h)

    @property
    def sigma(self):
        p = self.lib.fn("gauss_sigma")(self.h)
        return np.array([p[i] for i in range(self.dim)])

    @property
    def minisize(self):
        p = self.lib.fn("gauss_minisize")(self.h)
        return [int(p[i]) for i in range(self.dim)]

    @property
    def minisize_total(self):
        return int(self.lib.fn("gauss_minisize_total")(self.h))

    def bc_table(self, d, deriv):
        p = self.lib.fn("gauss_bc_table")(self.h, d, deriv)
        return np.ctypeslib.as_array(p, shape=(65536,)).copy()

    def _b(self, name, cast=float):
        p = self.lib.fn(name)(self.h)
        return np.array([cast(p[i]) for i in range(self.dim)])

    boundary_min = property(lambda s: s._b("gauss_boundary_min"))
    boundary_max = property(lambda s: s._b("gauss_boundary_max"))
    boundary_periodic = property(lambda s: s._b("gauss_boundary_periodic", int))

    def write(self, filename):
        self.lib.fn("gauss_write")(self.h, os.fsencode(filename))

    def multi_write(self, filename, lammps=0):
        self.lib.fn("gauss_multi_write")(self.h, os.fsencode(filename), lammps)


class Bias:
    def __init__(self, lib, config_path):
        self.lib = lib
        self.h = lib.fn("bias_create")(os.fsencode(config_path))
        self._mask = None

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.fn("bias_free")(self.h)
            self.h = None

    def setup(self, temperature, boltzmann):
        self.lib.fn("bias_setup")(self.h, temperature, boltzmann)

    def subdivide(self, sublo, subhi, boxlo, boxhi, periodic, skin):
        a, b, c, d, s = (_vec(v, 3) for v in (sublo, subhi, boxlo, boxhi, skin))
        p = _ivec(list(np.atleast_1d(periodic)) + [0, 0, 0])
        self.lib.fn("bias_subdivide")(self.h, _dp(a), _dp(b), _dp(c), _dp(d), p.ctypes.data_as(c_ip), _dp(s))

    def set_mask(self, mask):
        self._mask = _ivec(mask)
        self.lib.fn("bias_set_mask")(self.h, self._mask.ctypes.data_as(c_ip))

    def update_forces(self, positions, forces, apply_mask=-1):
        """positions, forces: float64 [n, stride] C-contiguous; forces updated in place."""
        assert positions.flags.c_contiguous and forces.flags.c_contiguous
        n, stride = positions.shape
        return self.lib.fn("bias_update_forces")(self.h, n, _dp(positions), _dp(forces), stride, apply_mask)

    def update_force(self, position):
        f = np.zeros(3)
        e = self.lib.fn("bias_update_force")(self.h, _dp(_vec(position, 3)), _dp(f))
        return e, f

    def add_hills(self, positions, runiform, apply_mask=-1):
        assert positions.flags.c_contiguous
        n, stride = positions.shape
        ru = _vec(runiform)
        self.lib.fn("bias_add_hills")(self.h, n, _dp(positions), stride, _dp(ru), apply_mask)

    def pre_add_hill(self, est):
        self.lib.fn("bias_pre_add_hill")(self.h, est)

    def add_hill(self, position, runiform):
        self.lib.fn("bias_add_hill")(self.h, _dp(_vec(position, 3)), runiform)

    def post_add_hill(self):
        self.lib.fn("bias_post_add_hill")(self.h)

    def pair_loop(self, r, second, runiform, hill_step, est):
        """One post_force of the reference's pair fix in its own order (fix_edm_pair.cpp:173-247):
        returns (energy, force[n], ncalls)."""
        r = _vec(r)
        sec = _ivec(second)
        ru = _vec(runiform if len(runiform) else [0.0])
        force = np.zeros(len(r))
        nc = np.zeros(1, dtype=np.int32)
        e = self.lib.fn("bias_pair_loop")(self.h, len(r), _dp(r), sec.ctypes.data_as(c_ip), _dp(ru), int(hill_step),
                                          int(est), _dp(force), nc.ctypes.data_as(c_ip))
        return e, force, int(nc[0])

    def write_bias(self, filename):
        self.lib.fn("bias_write_bias")(self.h, os.fsencode(filename))

    def write_lammps_table(self, filename):
        self.lib.fn("bias_write_lammps_table")(self.h, os.fsencode(filename))

    def write_histogram(self):
        self.lib.fn("bias_write_histogram")(self.h)

    def clear_histogram(self):
        self.lib.fn("bias_clear_histogram")(self.h)

    @property
    def gauss(self):
        return Gauss(self.lib, self.lib.fn("bias_gauss")(self.h), owned=False, keep=self)

    @property
    def hist(self):
        return Grid(self.lib, self.lib.fn("bias_hist")(self.h), owned=False, keep=self)

    def get(self, name):
        return self.lib.fn("bias_get")(self.h, name.encode())

    def set(self, name, value):
        self.lib.fn("bias_set")(self.h, name.encode(), float(value))

    def array(self, name):
        p = self.lib.fn("bias_array")(self.h, name.encode())
        return np.array([p[i] for i in range(int(self.get("dim")))])
