"""Child process of test_ball_list_equals_box_walk: per-hill integrals (the value add_value returns, gaussian_grid.h:227-281)
of seeded hills on 2-D / 3-D grids, periodic and walled, printed as hex floats.  The parent runs it with the list of
support offsets (default) and with EDM_HIP_TEST_FORCE=no_ball_list (the reference's stencil box walked point by point) and compares."""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import edm_amd.hip as H  # noqa: E402
import edm_amd.workloads as W  # noqa: E402


def main():
    H.require_gpu()
    out = {}
    cases = {
        # (lo, hi, spacing, periodic, sigma, boundary or None = the grid's own extent and periodicity)
        "p2": ([0, 0], [8, 8], [0.05, 0.04], [1, 1], [0.2, 0.15], None),
        "p3": ([0, 0, 0], [4, 4, 4], [0.1, 0.125, 0.1], [1, 1, 1], [0.2, 0.25, 0.15], None),
        "w2": ([0, 0], [6, 5], [0.05, 0.05], [0, 0], [0.2, 0.2], ([0.5, 0.4], [5.5, 4.6], [0, 0])),
        "w3": ([0, 0, 0], [3, 3, 3], [0.1, 0.1, 0.1], [0, 1, 0], [0.18, 0.2, 0.22], ([0.2, 0, 0.3], [2.8, 3, 2.7], [0, 1, 0])),
    }
    for name, (lo, hi, sp, per, sg, bnd) in cases.items():
        dim = len(lo)
        g = H.Gauss.create(lo, hi, sp, per, 1, sg)
        if bnd:
            g.set_boundary(*bnd)
        nh = 300
        x = W.uniform(4000 + dim, nh * dim).reshape(nh, dim) * (np.array(hi) - np.array(lo)) + np.array(lo)
        # some hills right at a wall / a periodic seam / a node
        x[0] = lo
        x[1] = np.array(hi) - 1e-9
        x[2] = np.array(lo) + np.array(sp) * 3
        h = 0.5 + W.uniform(4100 + dim, nh)
        a = g.hill_integrals(x, h)
        out[name] = [float(v).hex() for v in a]
        # ... and a list long enough for the wave-per-hill launch (more than 2048 hills)
        nl = 2600
        xl = W.uniform(4200 + dim, nl * dim).reshape(nl, dim) * (np.array(hi) - np.array(lo)) + np.array(lo)
        al = g.hill_integrals(xl, 0.75)
        out[name + "_long"] = [float(v).hex() for v in al]
        # batched add_value of a short list with its total (edm_hip_gauss_add_values): chained limiter launch, or
        # (EDM_HIP_TEST_FORCE=no_add_values_chain) integrals, sum, tile list and copy as launches of their own
        dx, dh, da = H.DeviceArray.from_host(x), H.DeviceArray.from_host(h), H.DeviceArray((nh,))
        tot = H.C.c_double(0)
        H.check(H.lib().edm_hip_gauss_add_values(g.h, nh, dx.ptr, dim, dh.ptr, 0.0, da.ptr, H.C.byref(tot)))
        gv, gd = g.download()
        out[name + "_addv"] = [float(v).hex() for v in da.to_host()]
        out[name + "_addv_total"] = [float(tot.value).hex()]
        out[name + "_grid"] = [hashlib.sha256(gv.tobytes() + gd.tobytes()).hexdigest()]
        del g
    print("INTEGRALS " + json.dumps(out))


if __name__ == "__main__":
    main()
