"""Worker of tests/test_gpu_pairfix_order.py::test_gate_wave_gives_up_...: three reference-order steps of the w1_density
scenario, a digest of energies and forces, and how often the second stream's gate wave gave up."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import edm_amd.hip as H
import pairfix_cases as PF

H.require_gpu()
workdir = sys.argv[1]
empty_first = len(sys.argv) > 2 and sys.argv[2] == "empty_first"   # the first step accepts no sample (uniforms of one)
name = "w1_density"
spec = PF.PAIRFIX[name]
cfg = os.path.join(workdir, "gate_%s_%d.edm" % (os.environ.get("EDM_HIP_TEST_FORCE") or "plain", empty_first))
with open(cfg, "w") as fh:
    fh.write(spec["cfg"] + "\nhills_filename %s.H\nhistogram_filename %s.hist\n" % (cfg, cfg))
b = H.Bias(cfg)
b.setup(1.0, 1.0)
b.subdivide([spec["lo"]], [spec["hi"]], [spec["lo"]], [spec["hi"]], [0], [spec["skin"]])
h = hashlib.sha256()
last = spec["nmax"]
for step in range(3):
    r, second, ru = PF.pairfix_inputs(name, step)
    xs, us = PF.staged_samples(r, second, ru)
    if empty_first and step == 0:
        us = np.ones_like(us)
    first = PF.first_calls(second)
    d_r, d_f = H.DeviceArray.from_host(r), H.DeviceArray.from_host(np.zeros(len(r)))
    d_first = H.DeviceArray.from_host(first)
    d_x, d_u = H.DeviceArray.from_host(xs), H.DeviceArray.from_host(us)
    e = b.pair_step_ordered_device(d_r, d_f, d_first, len(r), d_x, d_u, len(xs), est=last)
    h.update(np.float64(e).tobytes())
    h.update(d_f.to_host().tobytes())
    last = len(xs)
print("RESULT", h.hexdigest(), int(b.get("ord_gate_giveups")))
