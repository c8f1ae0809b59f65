"""Worker of tests/test_gpu_two_ranks.py::test_two_tenants_of_one_gpu: one single-rank bias object that runs
reference-order fix edm_pair steps back to back (record and force pass on a stream of their own, beside the hill batch's
launch) and prints a digest of everything it computed."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import edm_amd.hip as H
import edm_amd.workloads as W

tag, workdir, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
H.require_gpu()
cfg = os.path.join(workdir, "tenant_%s.edm" % tag)
with open(cfg, "w") as fh:
    fh.write("tempering 0\nhill_prefactor 0.5\nhill_density 250\ndimension 1\nbox_low 0\nbox_high 2.8\nbias_spacing 0.00025\n"
             "bias_sigma 0.025\nhills_filename %s/HILLS_%s\nhistogram_filename %s/HIST_%s\n" % (workdir, tag, workdir, tag))
b = H.Bias(cfg)
b.setup(1.0, 1.0)
b.subdivide([0], [2.8], [0], [2.8], [0], [0.3])
n = 262144
d_r = H.DeviceArray.from_host(W.pair_distances(n, 1))
d_first = H.DeviceArray.from_host(np.arange(n, dtype=np.int32))
d_f = H.DeviceArray.zeros((n,))
h = hashlib.sha256()
for step in range(steps):
    d_u = H.DeviceArray.from_host(W.uniform(100 + step % 7, n))
    e = b.pair_step_ordered_device(d_r, d_f, d_first, n, d_r, d_u, n, est=2 * n)
    h.update(np.float64(e).tobytes())
    if step % 16 == 0:
        h.update(d_f.to_host().tobytes())
v, dv = b.gauss.download()
h.update(v.tobytes())
h.update(dv.tobytes())
print("DIGEST", h.hexdigest(), int(b.get("hills_added")), int(b.get("poll_fallbacks")))
