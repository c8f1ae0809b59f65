"""Python face of the bias path with the reference package's names (python/edm/__init__.py:4-8,
python/edm/edm_python.cxx:8-17, python/edm/edm_bias_py.cpp): `EDMBias(input, T, kB)`, `set_box`,
`pre_add_hill` / `add_hill_r` / `post_add_hill`, `add_hill`, `get_force`, the writers.  The reference
builds this on Boost.Python (absent here); this one sits on the C ABI through `edm_amd.hip` (ctypes) and
runs on the GPU -- there is no CPU fallback."""
import random

import numpy as np

from . import hip as _H


class EDMBias:
    """edm_python.cxx:8 `EDMBias_Py(input_filename, temperature, boltzmann_constant)` (calls setup, edm_bias_py.cpp:26)."""

    def __init__(self, input_filename, temperature, boltzmann_constant):
        self._b = _H.Bias(input_filename)
        self._b.setup(temperature, boltzmann_constant)
        self.dim = int(self._b.get("dim"))

    def set_box(self, boxlo, boxhi, periodic):
        """edm_bias_py.cpp:31-53: subdivide(boxlo, boxhi, boxlo, boxhi, periodic, skin = 0).
        (The reference stores every flag into b_periodic[3] (:47), leaving the ones it passes on
        uninitialised; here the flags given are the flags used.)"""
        n = len(boxlo)
        lo = [float(v) for v in boxlo]
        hi = [float(v) for v in boxhi]
        self._b.subdivide(lo, hi, lo, hi, [int(p) for p in periodic][:n], [0.0] * n)

    def pre_add_hill(self, est_hill_count):
        self._b.pre_add_hill(int(est_hill_count))

    def add_hill_r(self, position, runiform):
        """edm_bias_py.cpp:56-66"""
        self._b.add_hill([float(v) for v in position][: self.dim], float(runiform))

    def post_add_hill(self):
        self._b.post_add_hill()

    def add_hill(self, position):
        """python/edm/__init__.py:5-8"""
        self.pre_add_hill(1)
        self.add_hill_r(position, random.random())
        self.post_add_hill()

    def get_force(self, position):
        """edm_bias_py.cpp:69-86: (bias energy, [dV/ds_j]) at one position (bias_->get_value_deriv)"""
        x = np.zeros((1, 3))
        x[0, : self.dim] = [float(v) for v in position][: self.dim]
        e, der = self._b.gauss.get_value_deriv(x)
        return float(e[0]), [float(v) for v in np.atleast_2d(der)[0][: self.dim]]

    def write_bias(self, filename):
        self._b.write_bias(filename)

    def write_lammps_table(self, filename):
        self._b.write_lammps_table(filename)

    def write_histogram(self):
        self._b.write_histogram()

    def clear_histogram(self):
        self._b.clear_histogram()

    # the public data members python callers of the reference read (edm_bias_py.cpp:22)
    @property
    def cum_bias(self):
        return self._b.get("cum_bias")

    @property
    def hill_prefactor(self):
        return self._b.get("hill_prefactor")
