"""Development aid: only the W3 / W4 coordinate-CV step loop of bench.py (fix edm, BASELINE configs[3] / [4]), for a kernel trace:
   rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/nd -o nd -- python3 tools/nd_steps.py 3"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import edm_amd.hip as H
import edm_amd.workloads as W

H.require_gpu()
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 3
c = W.C2D if dim == 2 else W.C3D
tmpdir = tempfile.mkdtemp()
natoms = 262144
x = W.atom_positions(natoms, 21 if dim == 2 else 31)
d_x = H.DeviceArray.from_host(x)
d_uu = H.DeviceArray.from_host(W.uniform(77, natoms))
cfgp = os.path.join(tmpdir, "nd.edm")
with open(cfgp, "w") as fh:
    fh.write("tempering 0\nhill_prefactor %g\nhill_density 250\n%sdimension %d\nbox_low %s\nbox_high %s\n"
             "bias_spacing %s\nbias_sigma %s\nhills_filename %s/HILLS\nhistogram_filename %s/HIST\n" % (
                 0.02 if dim == 3 else 0.5,
                 ("bias_per_step %s\n" % os.environ["ND_BIAS_PER_STEP"]) if os.environ.get("ND_BIAS_PER_STEP") else
                 ("bias_per_step 0.008\n" if dim == 3 else ""), dim,
                 " ".join("0" for _ in range(dim)), " ".join("64" for _ in range(dim)),
                 " ".join("%.10g" % v for v in c["spacing"]), " ".join("%.10g" % v for v in c["sigma"]), tmpdir, tmpdir))
bb = H.Bias(cfgp)
bb.setup(1.0, 1.0)
bb.subdivide([0.0] * dim, [64.0] * dim, [0.0] * dim, [64.0] * dim, [1] * dim, [0.0] * dim)
bb.set_hill_log(False)
d_fs = H.DeviceArray.zeros((natoms, 3))
for _ in range(3):
    bb.step_device(d_x, 3, d_fs, 3, natoms, d_uu, -1, natoms)
H.synchronize()
steps = int(os.environ.get("ND_STEPS", "40"))
t = time.perf_counter()
for _ in range(steps):
    bb.step_device(d_x, 3, d_fs, 3, natoms, d_uu, -1, natoms)
H.synchronize()
print("dim", dim, "ms_per_step", (time.perf_counter() - t) / steps * 1e3, "overflow_right", bb.get("overflow_right"), "hills", bb.get("hills_added"),
      "shared launches", bb.get("lookup_prep_launches"))
if os.environ.get("ND_HOST"):
    for k in range(int(os.environ.get("ND_PRE", "0"))):   # throwaway objects first: later streams get other copy engines
        pre = H.Bias(cfgp)
        pre.setup(1.0, 1.0)
        pre.subdivide([0.0] * dim, [64.0] * dim, [0.0] * dim, [64.0] * dim, [1] * dim, [0.0] * dim)
        pre.set_hill_log(False)
        xp = np.ascontiguousarray(x[:4096])
        pre.step_host(xp, np.zeros((4096, 3)), runiform=W.uniform(5, 4096), apply_mask=-1, hill_step=True, est=4096)
        del pre
    if os.environ.get("ND_PRE"):
        bb = H.Bias(cfgp)
        bb.setup(1.0, 1.0)
        bb.subdivide([0.0] * dim, [64.0] * dim, [0.0] * dim, [64.0] * dim, [1] * dim, [0.0] * dim)
    # the same loop from HOST arrays (edm_hip_bias_step_host), per-call times and the fallbacks of the polled completion
    xh = np.ascontiguousarray(x)
    fh = np.zeros((natoms, 3))
    uh = W.uniform(77, natoms)
    bb.set_hill_log(os.environ.get("ND_LOG", "1") == "1")
    for _ in range(3):
        bb.step_host(xh, fh, runiform=uh, apply_mask=-1, hill_step=True, est=natoms)
    pf0 = bb.get("poll_fallbacks")
    per = []
    for _ in range(steps):
        t0 = time.perf_counter()
        bb.step_host(xh, fh, runiform=uh, apply_mask=-1, hill_step=True, est=natoms)
        per.append((time.perf_counter() - t0) * 1e3)
    print("host step: mean %.3f ms  min %.3f  median %.3f  max %.3f  poll_fallbacks %d" % (
        sum(per) / len(per), min(per), sorted(per)[len(per) // 2], max(per), bb.get("poll_fallbacks") - pf0))
    print(" ".join("%.2f" % v for v in per))
