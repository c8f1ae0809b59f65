"""BASELINE configs[4] as a CONTROLLER step: 512^3 coordinate-CV bias grid, 262 144 atoms, hill_density 250,
bias_per_step small enough that every step crosses the limit (undo hill, deferred hills, overflow-buffer flush,
skipped rounds) -- through edm_hip_bias_step, i.e. the code only big grids reach (tile marking, launch sizing by
the expected count, culled gathers inside the fused step).

Oracle: the CPU restatement on the EQUIVALENT 64^3 periodic grid (same spacing and sigma, box [0, 8)^3) fed the
atoms at x mod 8.  On a periodic grid whose spacing is a power of two the per-hill integral depends only on the
position inside the cell and x mod 8 is exact, so every limiter decision (edm_bias.cpp:444-526, :313-380) must
come out identical: cum_bias, overflow indices, skip flag, hills_added, the HILLS event sequence.  The grids differ
(hills 8 apart fold onto each other in the small box) but folding is linear: the 8x8x8 images of the big grid,
summed, must equal the oracle's grid node for node.
"""
import numpy as np
import pytest

import edm_amd.hip as H
import edm_amd.workloads as W
from oracle import binding as B

pytestmark = pytest.mark.gpu

CFG = ("tempering 0\nhill_prefactor 0.02\nhill_density 250\nbias_per_step 0.008\ndimension 3\n"
       "box_low 0 0 0\nbox_high %g %g %g\nbias_spacing 0.125 0.125 0.125\nbias_sigma 0.25 0.25 0.25\n")
STATE = ("overflow_left", "overflow_right", "b_skip_hill_add", "hills_added", "steps")


def _events(path):
    out = []
    for line in open(path):
        t = line.split()
        out.append((int(t[0]), t[1], int(t[2])) + tuple(float(v) for v in t[3:]))
    return out


def test_w4_limited_steps_vs_equivalent_oracle(oracle_lib, workdir):
    H.require_gpu()
    paths = {}
    for tag, box in (("gpu", 64.0), ("ora", 8.0)):
        paths[tag] = str(workdir / (tag + ".edm"))
        open(paths[tag], "w").write(CFG % (box, box, box)
                                    + "hills_filename %s/HILLS_%s\nhistogram_filename %s/HIST_%s\n" % (workdir, tag, workdir, tag))
    b = H.Bias(paths["gpu"])
    o = B.Bias(oracle_lib, paths["ora"])
    b.setup(1.0, 1.0)
    o.setup(1.0, 1.0)
    b.subdivide([0] * 3, [64] * 3, [0] * 3, [64] * 3, [1, 1, 1], [0] * 3)
    o.subdivide([0] * 3, [8] * 3, [0] * 3, [8] * 3, [1, 1, 1], [0] * 3)
    assert list(b.gauss.number) == [512, 512, 512] and list(o.gauss.grid.number) == [64, 64, 64]
    n = 262144
    d_f = H.DeviceArray.zeros((n, 3))
    saw_skip = saw_buffered = False
    for step in range(6):
        x = W.atom_positions(n, 31 + step)
        u = W.uniform(931 + step, n)
        d_x = H.DeviceArray.from_host(x)
        d_u = H.DeviceArray.from_host(u)
        e = b.step_device(d_x, 3, d_f, 3, n, d_u)
        assert np.isfinite(e) and (step == 0) == (e == 0.0)
        o.add_hills(np.ascontiguousarray(np.mod(x, 8.0)), u)
        got = [b.get(k) for k in STATE]
        want = [o.get(k) for k in STATE]
        assert got == want, "limiter state after step %d: %s vs oracle %s" % (step, got, want)
        # bar 1e-6; the sums differ only by device exp() ulps and the fixed-order reduction of each integral
        assert abs(b.get("cum_bias") - o.get("cum_bias")) <= 1e-10 * o.get("cum_bias"), "cum_bias, step %d" % step
        saw_skip |= bool(got[2])
        saw_buffered |= got[0] != got[1]
    assert saw_skip and saw_buffered, "the workload must exercise the overflow buffer and a skipped round"
    H.synchronize()
    # CV histogram (bin width = bias_sigma, edm_bias.cpp:163): 256^3 bins, integer counts -- the folded images
    # must equal the oracle's 32^3 histogram exactly
    hv = b.hist.values
    assert hv.size == 256 ** 3 and o.hist.values.size == 32 ** 3
    assert np.array_equal(hv.reshape(8, 32, 8, 32, 8, 32).sum(axis=(0, 2, 4)).reshape(-1), o.hist.values)
    v, dv = b.gauss.download()
    del b
    # HILLS logs: the same events in the same order (type, hills_added), heights / bias_added to printed precision
    ge, oe = _events(str(workdir / "HILLS_gpu_0")), _events(str(workdir / "HILLS_ora_0"))
    assert len(ge) == len(oe) and len(ge) > 1000
    kinds = set()
    for a, w in zip(ge, oe):
        assert a[:3] == w[:3], (a, w)
        kinds.add(a[1])
        assert np.allclose(np.mod(a[3:6], 8.0), np.mod(w[3:6], 8.0), rtol=0, atol=2e-8), (a, w)
        assert abs(a[6] - w[6]) <= 2e-8 and abs(a[7] - w[7]) <= 2e-8, (a, w)
    assert {"h", "u", "b"} <= kinds, kinds
    # fold the 8^3 images of the big grid onto the small box
    ov, od = o.gauss.grid.values.reshape(64, 64, 64), o.gauss.grid.derivs.reshape(64, 64, 64, 3)
    fold = v.reshape(8, 64, 8, 64, 8, 64).sum(axis=(0, 2, 4))
    assert np.abs(fold - ov).max() <= 1e-10 * np.abs(ov).max(), "folded bias differs from the oracle's grid"
    foldd = dv.reshape(8, 64, 8, 64, 8, 64, 3).sum(axis=(0, 2, 4))
    assert np.abs(foldd - od).max() <= 1e-10 * np.abs(od).max(), "folded bias gradient differs from the oracle's grid"
    # locality on the big grid: bias only within the stencil reach of some logged hill position
    assert (v != 0).sum() < 6 * 300 * 23 ** 3


def test_w4_controller_with_a_narrow_gather_launch():
    """The culled gather's launch is capped at a few workgroups per CU and its workgroups stride over the tile list: the
    same controller test with the launch only SEVEN workgroups wide (EDM_HIP_TEST_FORCE=gather_wgs=7, read once per process), i.e.
    every workgroup walking hundreds of tiles, must pass unchanged."""
    import os
    import subprocess
    import sys

    if os.environ.get("EDM_TEST_CHILD"):
        pytest.skip("child run")
    env = dict(os.environ)
    env["EDM_HIP_TEST_FORCE"] = "gather_wgs=7"
    env["EDM_TEST_CHILD"] = "1"
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q"], env=env,
                         capture_output=True, text=True, timeout=900, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
