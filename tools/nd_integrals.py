"""Development aid: per-hill integrals of a long list on the 512^3 / 2048^2 grids (the wave-per-hill launch), timed.
   EDM_HIP_TEST_FORCE=no_ball_list python tools/nd_integrals.py 3   # the stencil box walked point by point"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import edm_amd.hip as H
import edm_amd.workloads as W

H.require_gpu()
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 3
c = W.C2D if dim == 2 else W.C3D
g = H.Gauss.create(c["lo"], c["hi"], c["spacing"], c["periodic"], 1, c["sigma"])
g.set_lookup_replica(0)
nh = 65536
x = W.atom_positions(nh, 11)[:, :dim].copy()
a = g.hill_integrals(x, 0.5)
H.synchronize()
t = time.perf_counter()
for _ in range(3):
    a = g.hill_integrals(x, 0.5)
H.synchronize()
print("dim", dim, "hills", nh, "ms per call (incl. upload)", (time.perf_counter() - t) / 3 * 1e3, "sum", float(a.sum()))
