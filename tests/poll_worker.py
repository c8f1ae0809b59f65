"""Child process of test_polled_completion_equals_stream_wait: one scripted run of hill-depositing steps with
calls that read the bias right behind them; prints a digest of everything a caller can observe.
The parent runs it twice -- EDM_HIP_POLL unset (results polled from host-mapped memory, the call returns while the
grid update still executes) and EDM_HIP_POLL=0 (every batch waits for its stream) -- and compares the digests."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import edm_amd.hip as H  # noqa: E402
import edm_amd.workloads as W  # noqa: E402


def main(workdir):
    H.require_gpu()
    dig = hashlib.sha256()

    def feed(*arrays):
        for a in arrays:
            dig.update(np.ascontiguousarray(np.asarray(a, dtype=np.float64)).tobytes())

    # 1-D pair CV with walls (boundary duplication chained behind the gather), limiter and overflow buffer busy
    cfg = os.path.join(workdir, "p1.edm")
    open(cfg, "w").write("tempering 0\nhill_prefactor 0.5\nhill_density 40\nbias_per_step 0.12\ndimension 1\nbox_low 0\n"
                         "box_high 2.8\nbias_spacing 0.001\nbias_sigma 0.05\nhills_filename %s/HILLS_p1\n"
                         "histogram_filename %s/HIST_p1\n" % (workdir, workdir))
    b = H.Bias(cfg)
    b.setup(1.0, 1.0)
    b.subdivide([0], [2.8], [0], [2.8], [0], [0.3])
    n = 6000
    probe = np.linspace(0.9, 2.75, 57).reshape(-1, 1)
    for step in range(8):
        r = W.pair_distances(n, 900 + step)
        d_r = H.DeviceArray.from_host(r)
        d_u = H.DeviceArray.from_host(W.uniform(950 + step, n))
        d_f = H.DeviceArray.zeros((n,))
        e = b.pair_step_device(d_r, d_f, n, d_r, d_u, n, est=n)
        # straight behind the step: lookups, the force array, and (every other step) the histogram
        v, dv = b.gauss.get_value_deriv(probe)
        feed([e], v, dv, d_f.to_host())
        if step % 2:
            feed(b.hist.values)
        if step == 5:
            b.clear_histogram()
    b.write_bias(os.path.join(workdir, "BIAS_p1"))
    gv, gd = b.gauss.download()
    feed(gv, gd, b.hist.values, [b.get("cum_bias"), b.get("overflow_right"), b.get("hills_added")])
    dig.update(open(os.path.join(workdir, "BIAS_p1"), "rb").read())
    dig.update(open(os.path.join(workdir, "HILLS_p1_0"), "rb").read())
    del b
    print("SECTION", 1, dig.hexdigest()[:16])

    # 2-D periodic coordinate CV through edm_hip_bias_step (flush + new hills every step)
    cfg = os.path.join(workdir, "p2.edm")
    open(cfg, "w").write("tempering 0\nhill_prefactor 0.5\nhill_density 60\nbias_per_step 0.2\ndimension 2\nbox_low 0 0\n"
                         "box_high 8 8\nbias_spacing 0.05 0.05\nbias_sigma 0.2 0.2\nhills_filename %s/HILLS_p2\n"
                         "histogram_filename %s/HIST_p2\n" % (workdir, workdir))
    b = H.Bias(cfg)
    b.setup(1.0, 1.0)
    b.subdivide([0, 0], [8, 8], [0, 0], [8, 8], [1, 1], [0, 0])
    na = 20000
    for step in range(5):
        x = W.uniform(1200 + step, 3 * na).reshape(na, 3) * 8.0
        d_x = H.DeviceArray.from_host(x)
        d_u = H.DeviceArray.from_host(W.uniform(1300 + step, na))
        d_f = H.DeviceArray.zeros((na, 3))
        e = b.step_device(d_x, 3, d_f, 3, na, d_u, -1, na)
        feed([e], d_f.to_host())
    gv, gd = b.gauss.download()
    feed(gv, gd, b.hist.values, [b.get("cum_bias"), b.get("overflow_right"), b.get("hills_added")])
    # (steps with an overflow flush: the force kernel rides in the launch that prepares the flush's hill list,
    #  EDM_HIP_TEST_FORCE=no_lookup_prep: a launch of its own, ahead of it)
    nshared = int(b.get("lookup_prep_launches"))
    del b
    # ... a 3-D grid with a group mask, device-drawn acceptance numbers, and the limiter far away (no flush: the force
    # kernel rides in the launch of the step's selection) for three steps, then binding (flushes carry it)
    for tag, limit, seed in (("p2m_free", 1000.0, 5), ("p2m_bound", 0.05, 6)):
        cfg = os.path.join(workdir, tag + ".edm")
        open(cfg, "w").write("tempering 0\nhill_prefactor 0.5\nhill_density 80\nbias_per_step %g\ndimension 3\nbox_low 0 0 0\n"
                             "box_high 6 6 6\nbias_spacing 0.1 0.1 0.1\nbias_sigma 0.2 0.2 0.2\nhills_filename %s/HILLS_%s\n"
                             "histogram_filename %s/HIST_%s\n" % (limit, workdir, tag, workdir, tag))
        b = H.Bias(cfg)
        b.setup(1.0, 1.0)
        b.subdivide([0, 0, 0], [6, 6, 6], [0, 0, 0], [6, 6, 6], [1, 1, 1], [0, 0, 0])
        na = 24000
        b.set_mask(1 + (np.arange(na) % 3 == 0).astype(np.int32))   # groups 1 and 2
        d_f = H.DeviceArray.zeros((na, 3))
        for step in range(4):
            x = W.uniform(1400 + 10 * seed + step, 3 * na).reshape(na, 3) * 6.0
            d_x = H.DeviceArray.from_host(x)
            d_u = H.DeviceArray.from_host(W.uniform(1450 + 10 * seed + step, na))
            e = b.step_device(d_x, 3, d_f, 3, na, d_u, 2, na)
            feed([e], d_f.to_host(), [b.get("cum_bias"), b.get("overflow_right"), b.get("hills_added")])
        gv, gd = b.gauss.download()
        feed(gv, gd, b.hist.values)
        nshared += int(b.get("lookup_prep_launches"))
        del b
    print("LOOKUP_PREP", nshared)
    print("SECTION", 2, dig.hexdigest()[:16])
    # 1-D pair CV, no HILLS log, limiter far away: the steps whose host call returns on the limiter's HEADER LINE alone
    # (EDM_HIP_TEST_FORCE=no_fast_header makes them wait for the completion word like every other polled batch)
    cfg = os.path.join(workdir, "p3.edm")
    open(cfg, "w").write("tempering 0\nhill_prefactor 0.5\nhill_density 120\nbias_per_step 50\ndimension 1\nbox_low 0\n"
                         "box_high 2.8\nbias_spacing 0.00025\nbias_sigma 0.025\nhills_filename %s/HILLS_p3\n"
                         "histogram_filename %s/HIST_p3\n" % (workdir, workdir))
    b = H.Bias(cfg)
    b.setup(1.0, 1.0)
    b.subdivide([0], [2.8], [0], [2.8], [0], [0.3])
    b.set_hill_log(False)
    n = 200_000
    for step in range(12):
        r = W.pair_distances(n, 1500 + step)
        d_r = H.DeviceArray.from_host(r)
        d_u = H.DeviceArray.from_host(W.uniform(1550 + step, n))
        d_f = H.DeviceArray.zeros((n,))
        e = b.pair_step_device(d_r, d_f, n, d_r, d_u, n, est=n)
        v, dv = b.gauss.get_value_deriv(probe)
        feed([e], v, dv, d_f.to_host(), [b.get("cum_bias"), b.get("hills_added"), b.get("overflow_right")])
    gv, gd = b.gauss.download()
    feed(gv, gd, b.hist.values)
    # forces-only calls in between (every step of a fix edm_pair run that deposits no hills): the host returns on the
    # workgroups' tagged partial sums instead of waiting for the stream -- short arrays and the LDS-staged long-array kernel
    for m, seed in ((n, 1600), (4001, 1601), (1_700_000, 1602)):
        rr = W.pair_distances(m, seed)
        d_rr = H.DeviceArray.from_host(rr)
        d_ff = H.DeviceArray.zeros((m,))
        for _ in range(3):
            e = b.pair_forces_device(d_rr, d_ff, m)
        feed([e], d_ff.to_host())
    print("POLLED_FORCES", int(b.get("polled_forces")))
    print("HEADER_RELEASES", int(b.get("header_releases")))
    del b
    print("SECTION", 3, dig.hexdigest()[:16])
    # 1-D pair CV whose walls lie INSIDE the rank's grid (a sub-domain with skin): hills at both walls meet boundary
    # corrections, and the duplication behind the gather copies the wall nodes outwards (gaussian_grid.h:571-630).  Only
    # the tiles near the walls take the ticket that decides it (EDM_HIP_TEST_FORCE=dup_ticket_all: every tile, the old way).
    for tag, limit in (("p4", 50.0), ("p5", 0.2)):
        cfg = os.path.join(workdir, tag + ".edm")
        open(cfg, "w").write("tempering 0\nhill_prefactor 0.5\nhill_density 150\nbias_per_step %g\ndimension 1\nbox_low 0.9\n"
                             "box_high 2.7\nbias_spacing 0.0005\nbias_sigma 0.03\nhills_filename %s/HILLS_%s\n"
                             "histogram_filename %s/HIST_%s\n" % (limit, workdir, tag, workdir, tag))
        b = H.Bias(cfg)
        b.setup(1.0, 1.0)
        b.subdivide([0.5], [3.0], [0.0], [3.5], [0], [0.3])
        b.set_hill_log(tag == "p5")
        n = 120_000
        for step in range(6):
            r = W.pair_distances(n, 1700 + step)
            d_r = H.DeviceArray.from_host(r)
            d_u = H.DeviceArray.from_host(W.uniform(1750 + step, n))
            d_f = H.DeviceArray.zeros((n,))
            e = b.pair_step_device(d_r, d_f, n, d_r, d_u, n, est=n)
            feed([e], d_f.to_host(), [b.get("cum_bias"), b.get("hills_added"), b.get("overflow_right")])
        gv, gd = b.gauss.download()
        feed(gv, gd, b.hist.values)
        geo = b.gauss.geometry
        print("WALLS_INSIDE", tag, float(geo.min[0]), float(geo.max[0]), float(np.abs(gv).max()))
        del b
        print("SECTION", tag, dig.hexdigest()[:16])
    print("SECTION", 4, dig.hexdigest()[:16])
    print("DIGEST", dig.hexdigest())


if __name__ == "__main__":
    main(sys.argv[1])
