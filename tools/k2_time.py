"""Development aid: K2 (coordinate-CV lookup on the replica) by its dispatch timestamps, 262 144 and 2 097 152 atoms,
2048^2 and 512^3 grids:   python tools/k2_time.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import edm_amd.hip as H
import edm_amd.workloads as W

H.require_gpu()
for tag, c, per_atom in (("2048^2", W.C2D, 156), ("512^3", W.C3D, 332)):
    gg = H.Gauss.create(c["lo"], c["hi"], c["spacing"], c["periodic"], 1, c["sigma"])
    x0 = W.atom_positions(250, 5)
    hills = H.DeviceArray.from_host(np.ascontiguousarray(x0))
    tot = H.C.c_double(0)
    H.check(H.lib().edm_hip_gauss_add_values(gg.h, 250, hills.ptr, 3, None, 0.01, None, H.C.byref(tot)))
    for natoms in (262144, 2097152):
        x = W.atom_positions(natoms, 21)
        d_x = H.DeviceArray.from_host(x)
        d_f = H.DeviceArray.zeros((natoms, 3))
        e = H.C.c_double(0)
        H.check(H.lib().edm_hip_gauss_update_forces(gg.h, natoms, d_x.ptr, 3, d_f.ptr, 3, None, -1, H.C.byref(e)))
        gg.profile_enable(True)
        gg.profile_read(reset=True)
        for _ in range(20):
            H.check(H.lib().edm_hip_gauss_update_forces(gg.h, natoms, d_x.ptr, 3, d_f.ptr, 3, None, -1, H.C.byref(e)))
        ms, ln = gg.profile_read(reset=True)
        gg.profile_enable(False)
        us = ms / ln * 1e3
        print("%s %8d atoms: %7.2f us  %.3f of 8 TB/s (%.2f TB/s algorithmic)  E=%.10g" % (
            tag, natoms, us, per_atom * natoms / (us * 1e-6) / 8e12, per_atom * natoms / (us * 1e-6) / 1e12, e.value))
    del gg
