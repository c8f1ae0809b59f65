"""Development aid: the window form of the reference-order force pass (k_pair_forces_ordered_win) on the 38.8 M pairs of W2,
with the pairs' first add_hill calls spread over the step's samples (every workgroup's run crosses ~0.5 hills) or all
behind the last sample (one window per workgroup)."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import edm_amd.hip as H
import edm_amd.workloads as W
from bench import make_bias

H.require_gpu()
tmpdir = tempfile.mkdtemp()
npairs, n2 = 1 << 20, W.W2_PAIRS
b = H.Bias(make_bias(H, tmpdir, "gpu", 0))
b.setup(1.0, 1.0)
b.subdivide([0], [2.8], [0], [2.8], [0], [0.3])
g = b.gauss
d_r = H.DeviceArray.from_host(W.pair_distances(npairs, 1))
d_u = H.DeviceArray.from_host(W.uniform(3, npairs))
d_r2 = H.DeviceArray.from_host(W.pair_distances(n2, 11))
d_f2 = H.DeviceArray((n2,))
for mode in ("spread", "all_behind"):
    first = (np.arange(n2, dtype=np.int64) * npairs // n2).astype(np.int32) if mode == "spread" else np.full(n2, npairs, dtype=np.int32)
    d_first2 = H.DeviceArray.from_host(first)
    for _ in range(2):
        b.pair_step_ordered_device(d_r2, d_f2, d_first2, n2, d_r, d_u, npairs, 2 * npairs)
    g.profile_enable(True)
    g.profile_read(reset=True)
    for _ in range(6):
        b.pair_step_ordered_device(d_r2, d_f2, d_first2, n2, d_r, d_u, npairs, 2 * npairs)
    ms, l = g.profile_read(reset=True)
    g.profile_enable(False)
    print(mode, "kernel_us", ms / l * 1e3)
g.pair_forces_device(d_r2, d_f2, n2)
g.profile_enable(True)
g.profile_read(reset=True)
for _ in range(6):
    g.pair_forces_device(d_r2, d_f2, n2)
ms, l = g.profile_read(reset=True)
print("K1 kernel_us", ms / l * 1e3)
