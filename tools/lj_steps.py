"""Development aid: only the 32k-atom LJ-melt-from-positions loop of bench.py (fix edm_pair gpu_list), for a kernel trace:
   rocprofv3 --kernel-trace --stats -d gpurun_out/lj -- python3 tools/lj_steps.py"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import edm_amd.hip as H
import edm_amd.workloads as W
from bench import make_bias
from scipy.spatial import cKDTree

H.require_gpu()
tmpdir = tempfile.mkdtemp()
na = 32000
box = (na / 0.8442) ** (1.0 / 3.0)
xa = W.uniform(5, 3 * na).reshape(na, 3) * box
pr = cKDTree(xa).query_pairs(2.8, output_type="ndarray")
pr = pr[np.lexsort((pr[:, 1], pr[:, 0]))].astype(np.int32)
bl = H.Bias(make_bias(H, tmpdir, "lj", 0))
bl.setup(1.0, 1.0)
bl.subdivide([0], [2.8], [0], [2.8], [0], [0.3])
bl.set_hill_log(False)
bl.set_device_rng(True, 777)
if os.environ.get("LJ_ORDER", "0") == "1":   # the reference fix's order (the rewritten fix's default)
    bl.set("reference_order", 1)
bl.pair_list_upload(pr[:, 0], pr[:, 1], np.ones(na, dtype=np.int32))
d_xa = H.DeviceArray.from_host(xa)
d_fa = H.DeviceArray.zeros((na, 3))
calls = 2 * len(pr)
for _ in range(3):
    _, calls = bl.pair_list_step_device(na, 1, 1, d_xa, d_fa, True, calls)
H.synchronize()
steps = int(os.environ.get("LJ_STEPS", "100"))
t = time.perf_counter()
for _ in range(steps):
    _, calls = bl.pair_list_step_device(na, 1, 1, d_xa, d_fa, True, calls)
H.synchronize()
print("ms_per_step", (time.perf_counter() - t) / steps * 1e3, "overflow_right", bl.get("overflow_right"), "hills", bl.get("hills_added"))
