"""Worker of tests/test_gpu_pairfix_order.py::test_long_pair_arrays_take_the_lds_window_form: two reference-order steps
on a pair array long enough for the LDS-window force pass (k_pair_forces_ordered_win), or -- with
EDM_HIP_TEST_FORCE=no_k1o_window in the environment -- for the short-array kernel on the same array."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import edm_amd.hip as H
import edm_amd.workloads as W

workdir, shuffled = sys.argv[1], len(sys.argv) > 2 and sys.argv[2] == "shuffled"
H.require_gpu()
cfg = os.path.join(workdir, "win_%s_%d.edm" % (os.environ.get("EDM_HIP_TEST_FORCE") or "window", shuffled))
with open(cfg, "w") as fh:
    fh.write("tempering 0\nhill_prefactor 0.5\nhill_density 250\ndimension 1\nbox_low 0.2\nbox_high 2.7\nbias_spacing 0.00025\n"
             "bias_sigma 0.025\nhills_filename %s.H\nhistogram_filename %s.hist\n" % (cfg, cfg))
b = H.Bias(cfg)
b.setup(1.0, 1.0)
b.subdivide([0.0], [2.8], [0.0], [2.8], [0], [0.3])    # (walls strictly inside the grid: outward copy nodes exist)
n, ns = 2000003, 300000
r = W.pair_distances(n, 5)
r[::1001] = 0.1                                        # (some pairs below the window / outside the walls)
first = (np.arange(n, dtype=np.int64) * ns // n).astype(np.int32)
if shuffled:                                           # (an array that does not ascend: every pair down the general form)
    perm = np.random.default_rng(3).permutation(n)
    r, first = np.ascontiguousarray(r[perm]), np.ascontiguousarray(first[perm])
d_r, d_first, d_f = H.DeviceArray.from_host(r), H.DeviceArray.from_host(first), H.DeviceArray.zeros((n,))
d_s = H.DeviceArray.from_host(W.pair_distances(ns, 6))
h = hashlib.sha256()
energies = []
for step in range(2):
    d_u = H.DeviceArray.from_host(W.uniform(40 + step, ns))
    e = b.pair_step_ordered_device(d_r, d_f, d_first, n, d_s, d_u, ns, est=2 * ns)
    energies.append(e)
    f = d_f.to_host()
    if shuffled:
        out = np.empty_like(f)
        out[perm] = f
        f = out
    h.update(f.tobytes())
print("RESULT", h.hexdigest(), " ".join("%.17g" % e for e in energies), int(b.get("hills_added")))
