"""Development aid: the p3 section of tests/poll_worker.py alone (1-D pair CV, no HILLS log, limiter far away, values read
straight behind every step), digest per run -- run-to-run differences mean a race between the host's release and the step."""
import hashlib
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import edm_amd.hip as H
import edm_amd.workloads as W

H.require_gpu()
what = sys.argv[1] if len(sys.argv) > 1 else "all"
for rep in range(4):
    workdir = tempfile.mkdtemp()
    digs = {k: hashlib.sha256() for k in ("e", "v", "f", "st", "grid")}
    cfg = os.path.join(workdir, "p3.edm")
    open(cfg, "w").write("tempering 0\nhill_prefactor 0.5\nhill_density 120\nbias_per_step 50\ndimension 1\nbox_low 0\n"
                         "box_high 2.8\nbias_spacing 0.00025\nbias_sigma 0.025\nhills_filename %s/HILLS_p3\n"
                         "histogram_filename %s/HIST_p3\n" % (workdir, workdir))
    b = H.Bias(cfg)
    b.setup(1.0, 1.0)
    b.subdivide([0], [2.8], [0], [2.8], [0], [0.3])
    b.set_hill_log(False)
    n = 200_000
    probe = np.linspace(0.9, 2.75, 57).reshape(-1, 1)
    for step in range(12):
        r = W.pair_distances(n, 1500 + step)
        d_r = H.DeviceArray.from_host(r)
        d_u = H.DeviceArray.from_host(W.uniform(1550 + step, n))
        d_f = H.DeviceArray.zeros((n,))
        e = b.pair_step_device(d_r, d_f, n, d_r, d_u, n, est=n)
        digs["e"].update(np.float64(e).tobytes())
        if what in ("all", "v"):
            v, dv = b.gauss.get_value_deriv(probe)
            digs["v"].update(v.tobytes() + dv.tobytes())
        if what in ("all", "f"):
            digs["f"].update(d_f.to_host().tobytes())
        digs["st"].update(np.array([b.get("cum_bias"), b.get("hills_added"), b.get("overflow_right")]).tobytes())
    gv, gd = b.gauss.download()
    digs["grid"].update(gv.tobytes() + gd.tobytes() + np.asarray(b.hist.values, dtype=np.float64).tobytes())
    print(rep, " ".join("%s=%s" % (k, d.hexdigest()[:10]) for k, d in digs.items()), "fused", b.get("fused_steps"))
    del b
