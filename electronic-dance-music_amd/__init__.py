"""MI355X-native implementation of the EDM per-timestep bias hot path.

The product is the C-ABI library ``csrc/libedm_hip.so`` (include/edm_hip.h) and
the C++ class layer ``libedm.so`` (include/edm/*.h); this Python package is the
thin ctypes mirror of the reference's operator interface used by tests and
bench.py.  It never falls back to a CPU path: importing ``edm_amd.hip`` without
the built HIP library raises.
"""
__all__ = ["workloads"]
