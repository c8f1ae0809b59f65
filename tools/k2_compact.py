"""K2 on the node records (replica off) vs on the lookup replica, 262144 random atoms: kernel microseconds."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import edm_amd.hip as H
import edm_amd.workloads as W
for c in (W.C2D, W.C3D):
    for mode in (0, -1):
        g = H.Gauss.create(c["lo"], c["hi"], c["spacing"], c["periodic"], 1, c["sigma"])
        g.set_lookup_replica(mode)
        n = 262144
        x = W.atom_positions(n, 31)
        d_x = H.DeviceArray.from_host(x); d_f = H.DeviceArray.zeros((n, 3))
        g.add_values(x[:250].copy(), 0.01)
        e = H.C.c_double(0)
        H.check(H.lib().edm_hip_gauss_update_forces(g.h, n, d_x.ptr, 3, d_f.ptr, 3, None, -1, H.C.byref(e)))
        g.profile_enable(True); g.profile_read(reset=True)
        for _ in range(20):
            H.check(H.lib().edm_hip_gauss_update_forces(g.h, n, d_x.ptr, 3, d_f.ptr, 3, None, -1, H.C.byref(e)))
        ms, ln = g.profile_read(reset=True)
        print("dim %d replica %s: %.1f us" % (c["dim"], "auto" if mode else "off", ms / ln * 1e3))
        del g
