"""MI355X-native implementation of the EDM per-timestep bias hot path.

The product is the C-ABI library ``csrc/libedm_hip.so`` (include/edm_hip.h) and
the C++ class layer over it (include/edm/*.h); this Python package is the thin
host mirror used by tests and bench.py:

  hip        ctypes binding of the C ABI (GaussGrid / EDMBias operator interface)
  workloads  deterministic synthetic inputs
  parallel   torch.distributed mirror of the multi-GPU hill-exchange protocol

There is no CPU fallback: ``hip`` raises if the HIP library is missing or no GPU is visible.
"""
__all__ = ["hip", "workloads", "parallel"]
