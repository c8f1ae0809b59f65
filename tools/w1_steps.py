"""Development aid: only the W1 step loop of bench.py (fix edm_pair, BASELINE configs[1]), for a kernel trace:
   W1_ORDER=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/w1 -o w1 -- python3 tools/w1_steps.py"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import edm_amd.hip as H
import edm_amd.workloads as W
from bench import make_bias

H.require_gpu()
tmpdir = tempfile.mkdtemp()
npairs = 1 << 20
b = H.Bias(make_bias(H, tmpdir, "gpu", 0))
b.setup(1.0, 1.0)
b.subdivide([0], [2.8], [0], [2.8], [0], [0.3])
b.set_hill_log(os.environ.get("W1_LOG", "0") == "1")   # W1_LOG=1: the reference's per-hill HILLS log (the library's default)
hills0 = np.zeros((4096, 1))
hills0[:, 0] = W.pair_distances(4096, 2)
b.gauss.add_values(hills0, 1e-3)
d_r = H.DeviceArray.from_host(W.pair_distances(npairs, 1))
d_u = H.DeviceArray.from_host(W.uniform(3, npairs))
d_f = H.DeviceArray.zeros((npairs,))
est = 2 * npairs
# W1_ORDER=1: the step in the reference's order (edm_hip_bias_pair_step_ordered; one add_hill sample per pair here)
ordered = os.environ.get("W1_ORDER", "0") == "1"
d_first = H.DeviceArray.from_host(np.arange(npairs, dtype=np.int32))


def step():
    if ordered:
        return b.pair_step_ordered_device(d_r, d_f, d_first, npairs, d_r, d_u, npairs, est)
    return b.pair_step_device(d_r, d_f, npairs, d_r, d_u, npairs, est)


for _ in range(10):
    step()
H.synchronize()
steps = int(os.environ.get("W1_STEPS", "200"))
t = time.perf_counter()
for _ in range(steps):
    step()
H.synchronize()
print("ordered" if ordered else "batch", "ms_per_step", (time.perf_counter() - t) / steps * 1e3,
      "hills_added", b.get("hills_added"), "gate give-ups", b.get("ord_gate_giveups"))
