"""Edge cases of the hot path on the GPU, through the C ABI: empty and odd-sized batches, samples that sit
exactly on grid nodes / boundaries (the K1 fast path's exact-index fallback), fully masked batches, hills
that the boundary rejects, overflow-buffer exhaustion and call-order errors."""
import numpy as np
import pytest

import edm_amd.hip as H
import edm_amd.workloads as W
from oracle import binding as B

pytestmark = pytest.mark.gpu

C1D = dict(lo=[0.0], hi=[2.8], sp=[0.00025], per=[0], sg=[0.025])


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    H.require_gpu()
    yield


def _pair(oracle_lib, c=C1D, nh=200):
    g = H.Gauss.create(c["lo"], c["hi"], c["sp"], c["per"], 1, c["sg"])
    o = B.Gauss.create(oracle_lib, c["lo"], c["hi"], c["sp"], c["per"], 1, c["sg"])
    hx = np.zeros((nh, 3))
    hx[:, 0] = W.pair_distances(nh, 5)
    g.add_values(hx, 0.01)
    for x in hx:
        o.add_value(x[:1], 0.01)
    return g, o


def test_empty_and_tiny_batches(oracle_lib):
    g, o = _pair(oracle_lib)
    e, f = g.pair_forces(np.zeros(0))
    assert e == 0.0 and f.size == 0
    assert g.update_forces(np.zeros((0, 3)), np.zeros((0, 3))) == 0.0
    assert g.add_values(np.zeros((0, 3)), 1.0).size == 0
    for n in (1, 2, 3, 63, 64, 65, 1025):  # odd sizes exercise the tail of the vectorised kernel
        r = W.pair_distances(n, 40 + n)
        e, f = g.pair_forces(r)
        ref = [o.get_value_deriv([x]) for x in r]
        assert abs(e - sum(v for v, _ in ref)) <= 1e-10 * max(abs(e), 1e-30)
        ref_f = np.array([-d[0] for _, d in ref])
        # dV/dr is a difference of O(V/dx) terms: ~1e-12 relative-to-scale rounding noise is inherent
        assert np.allclose(f, ref_f, rtol=1e-9, atol=1e-11 * np.abs(ref_f).max())


def test_samples_on_nodes_and_boundaries(oracle_lib):
    """x = k*dx exactly (and one ulp either side): the node index must equal the reference's
    floor((x-min)/dx) bit for bit, in the fast pair kernel and in the generic kernel."""
    g, o = _pair(oracle_lib)
    og = o.grid
    dx = float(og.dx[0])
    ks = np.concatenate([np.arange(0, 11201, 37), [0, 1, 11198, 11199, 11200]]).astype(np.float64)
    base = ks * dx
    r = np.concatenate([base, np.nextafter(base, 10), np.nextafter(base, -10), [2.8, 2.8 - 1e-16, 0.0, -0.0, 2.8 + dx, -1e-300]])
    idx = g.sample_index(r.reshape(-1, 1))
    want = np.array([og.multi2one(og.get_index([x])) if (o.in_bounds([x]) and og.in_grid([x])) else -1 for x in r])
    assert np.array_equal(idx, want)
    e, f = g.pair_forces(r)
    ref = [o.get_value_deriv([x]) for x in r]
    ref_f = np.array([-d[0] for _, d in ref])
    assert np.allclose(f, ref_f, rtol=1e-9, atol=1e-11 * np.abs(ref_f).max())
    assert abs(e - sum(v for v, _ in ref)) <= 1e-10 * abs(e)
    # big batch -> the LDS-window variant of the kernel; same node-aligned samples repeated
    big = np.tile(r, (1 << 22) // r.size + 1)
    e2, f2 = g.pair_forces(big)
    assert np.allclose(f2[: r.size], ref_f, rtol=1e-9, atol=1e-11 * np.abs(ref_f).max())
    assert np.array_equal(f2[: r.size], f2[r.size: 2 * r.size])


def test_fully_masked_and_out_of_range():
    g = H.Gauss.create(C1D["lo"], C1D["hi"], C1D["sp"], C1D["per"], 1, C1D["sg"])
    g.add_values(np.array([[1.5, 0, 0]]), 1.0)
    n = 1000
    x = np.zeros((n, 3))
    x[:, 0] = W.pair_distances(n, 3)
    f = np.ones((n, 3))
    e = g.update_forces(x, f, mask=np.zeros(n, dtype=np.int32), apply_mask=1)
    assert e == 0.0 and np.array_equal(f, np.ones((n, 3)))
    far = np.full(n, 7.5)  # outside boundary and grid: contributes (0, 0) (gaussian_grid.h:128-135)
    e, fr = g.pair_forces(far)
    assert e == 0.0 and not fr.any()
    added = g.add_values(np.array([[7.5, 0, 0], [-0.1, 0, 0]]), 1.0)  # rejected by the boundary (:214-216)
    assert not added.any()


def _bias(tmp_path, tag, text):
    cfg = str(tmp_path / (tag + ".edm"))
    open(cfg, "w").write(text + "hills_filename %s/H_%s\nhistogram_filename %s/HIST_%s\n" % (tmp_path, tag, tmp_path, tag))
    b = H.Bias(cfg)
    b.setup(1.0, 1.0)
    b.subdivide([0], [2.8], [0], [2.8], [0], [0.3])
    return b


BASE = "tempering 0\ndimension 1\nbox_low 0\nbox_high 2.8\nbias_spacing 0.001\nbias_sigma 0.05\n"


def test_overflow_buffer_exhaustion_is_reported(tmp_path):
    """More deferred hills than the 2048-slot overflow buffer holds: the reference aborts
    (edm_bias.cpp:501-507); the C ABI returns EDM_HIP_ERR_OVERFLOW."""
    b = _bias(tmp_path, "ovf", BASE + "hill_prefactor 1.0\nbias_per_step 0.0001\n")
    b.set_hill_log(False)
    n = 3000
    r = W.pair_distances(n, 77).reshape(-1, 1)
    with pytest.raises(H.EdmHipError) as ei:
        b.add_hills(r, np.ones(n))
    assert "overflow buffer is full" in str(ei.value)


def test_call_order_errors(tmp_path):
    b = _bias(tmp_path, "ord", BASE + "hill_prefactor 0.5\n")
    with pytest.raises(H.EdmHipError):
        b.add_hill([1.0], 0.5)  # before pre_add_hill (edm_bias.cpp:530-531)
    b.pre_add_hill(4)
    b.add_hill([1.0], 0.5)
    b.post_add_hill()
    assert b.get("cum_bias") > 0 and b.get("steps") == 1


def test_skipped_round_when_buffer_not_drained(tmp_path, oracle_lib):
    """Leftover buffered bias makes the whole next round of new hills be skipped (edm_bias.cpp:434-439)."""
    text = BASE + "hill_prefactor 1.0\nbias_per_step 0.3\n"
    b = _bias(tmp_path, "skip", text)
    cfg_o = str(tmp_path / "skip_o.edm")
    open(cfg_o, "w").write(text + "hills_filename %s/H_o\nhistogram_filename %s/HIST_o\n" % (tmp_path, tmp_path))
    o = B.Bias(oracle_lib, cfg_o)
    o.setup(1.0, 1.0)
    o.subdivide([0], [2.8], [0], [2.8], [0], [0.3])
    for step in range(6):
        r = W.pair_distances(5, 600 + step).reshape(-1, 1)
        u = np.ones(5)
        b.pre_add_hill(1)
        o.pre_add_hill(1)
        for x in r:
            b.add_hill(x, 1.0)
            o.add_hill(x, 1.0)
        b.post_add_hill()
        o.post_add_hill()
        keys = ("overflow_left", "overflow_right", "b_skip_hill_add", "hills_added", "steps")
        assert [b.get(k) for k in keys] == [o.get(k) for k in keys], step
        assert abs(b.get("cum_bias") - o.get("cum_bias")) <= 1e-10 * o.get("cum_bias")
    assert np.array_equal(b.hist.values, o.hist.values)


@pytest.mark.parametrize("entry", ["add_hills", "pair_step", "communicator", "pair_step_ordered", "communicator_ordered"])
def test_deferred_bound_exceeded_redo(entry, tmp_path, oracle_lib):
    """A stochastic step is queued against a launch bound (4 x the expected count + 128) before the accepted count
    is known; when the count exceeds the bound the limiter flags it (error 2), the chained gather / histogram /
    read-back / tile-flag clean-up must all stand down, and the controller redoes the hill path synchronously with
    the same device-RNG cycle.  Driven here by uniforms that accept EVERY sample (3000 against a bound of 256),
    through add_hills, the fused pair_step (forces ride in the aborted launch and must stay valid) and the packed
    exchange of a one-rank communicator -- and the reference-order step, whose records and force pass are queued BEHIND
    the aborted batch before the host knows (they count zero hills, and the redo launches them again).  Bit for bit
    against a handle that never defers (debug_force_sync), at the
    parity tolerance against the oracle, and with ordinary steps before and after (tickets / flags left clean)."""
    text = BASE + "hill_prefactor 0.5\nhill_density 20\nbias_per_step 50\n"
    n = 3000
    runs = {}
    for tag in ("deferred", "sync"):
        b = _bias(tmp_path, entry + "_" + tag, text)
        if entry.startswith("communicator"):
            b.comm_init(H.comm_unique_id(), 1, 0)
        if tag == "sync":
            b.set("debug_force_sync", 1)
        out = []
        for step in range(4):
            r = W.pair_distances(n, 880 + step)
            u = np.zeros(n) if step == 1 else W.uniform(890 + step, n)   # step 1: everything accepted
            d_r = H.DeviceArray.from_host(r)
            d_u = H.DeviceArray.from_host(u)
            d_f = H.DeviceArray.zeros((n,))
            if entry == "pair_step":
                e = b.pair_step_device(d_r, d_f, n, d_r, d_u, n, est=n)
            elif entry.endswith("_ordered"):
                d_first = H.DeviceArray.from_host(np.arange(n, dtype=np.int32))
                e = b.pair_step_ordered_device(d_r, d_f, d_first, n, d_r, d_u, n, est=n)
            else:
                e = b.pair_forces_device(d_r, d_f, n)
                b.add_hills_device(d_r, n, 1, d_u, -1, est=n)
            out.append((e, d_f.to_host(), [b.get(k) for k in ("overflow_left", "overflow_right", "b_skip_hill_add",
                                                              "hills_added", "steps", "cum_bias")]))
        v, dv = b.gauss.download()
        runs[tag] = (out, v, dv, b.hist.values, b.get("bound_redos"))
        del b
    assert runs["deferred"][4] == 1 and runs["sync"][4] == 0, "exactly the all-accepted step must take the redo path"
    for (e1, f1, s1), (e2, f2, s2) in zip(runs["deferred"][0], runs["sync"][0]):
        assert e1 == e2 and np.array_equal(f1, f2) and s1 == s2
    for k in (1, 2, 3):
        assert np.array_equal(runs["deferred"][k], runs["sync"][k])
    assert open(str(tmp_path / ("H_%s_deferred_0" % entry))).read() == open(str(tmp_path / ("H_%s_sync_0" % entry))).read()
    # ... and against the oracle (fix_edm_pair order: pre_add_hill, forces, add_hill per sample, post_add_hill)
    cfg_o = str(tmp_path / (entry + "_o.edm"))
    open(cfg_o, "w").write(text + "hills_filename %s/H_o_%s\nhistogram_filename %s/HIST_o_%s\n" % (tmp_path, entry, tmp_path, entry))
    o = B.Bias(oracle_lib, cfg_o)
    o.setup(1.0, 1.0)
    o.subdivide([0], [2.8], [0], [2.8], [0], [0.3])
    for step in range(4):
        r = W.pair_distances(n, 880 + step)
        u = np.zeros(n) if step == 1 else W.uniform(890 + step, n)
        fo = np.zeros((n, 1))
        if entry.endswith("_ordered"):   # the reference fix's own loop: update_force, then add_hill, pair after pair
            eo, f_loop, _ = o.pair_loop(r, np.zeros(n, dtype=np.int32), u, 1, n)
            fo[:, 0] = f_loop
        else:
            if entry == "pair_step":
                o.pre_add_hill(n)
                eo = o.update_forces(r.reshape(-1, 1).copy(), fo)
            else:
                eo = o.update_forces(r.reshape(-1, 1).copy(), fo)
                o.pre_add_hill(n)
            for i in np.nonzero(u < 20.0 / n)[0]:
                o.add_hill([r[i]], float(u[i]))
            o.post_add_hill()
        e1, f1, s1 = runs["deferred"][0][step]
        assert abs(e1 - eo) <= 1e-9 * max(abs(eo), 1e-300)
        assert np.allclose(f1, fo[:, 0], rtol=1e-8, atol=1e-11 * max(np.abs(fo).max(), 1e-300))
        assert s1[:5] == [o.get(k) for k in ("overflow_left", "overflow_right", "b_skip_hill_add", "hills_added", "steps")]
        assert abs(s1[5] - o.get("cum_bias")) <= 1e-10 * o.get("cum_bias")
    ov = o.gauss.grid.values
    assert np.allclose(runs["deferred"][1], ov, rtol=1e-9, atol=1e-13 * np.abs(ov).max())
    assert np.array_equal(runs["deferred"][3], o.hist.values)
    assert runs["deferred"][0][1][2][1] > 0, "the all-accepted step must have pushed hills into the overflow buffer"


# ---------------------------------------------------------------------------------
# k_integrals_gather: which half of the launch is dispatched first (gather tiles that WAIT for the limiter's word, or the
# per-hill integrals the limiter needs).  The tiles go first only where they and the expected hills fit the kernel's
# measured residency (hipOccupancyMaxActiveBlocksPerMultiprocessor of the launched instantiation) with room to spare,
# and never on a device another rank uses; either order must give the same bits.
# ---------------------------------------------------------------------------------
@pytest.mark.parametrize("tiles", [690, 705, 740, 1500], ids=lambda t: "%d_tiles" % t)
def test_integrals_gather_dispatch_order_near_the_residency_limit(tiles, workdir, oracle_lib):
    import edm_amd.workloads as W

    nodes = 32 * tiles - 5                      # 32-node gather tiles; 3 workgroups per CU x 256 CUs = 768 slots
    hi = 2.8
    spacing = hi / (nodes - 1)
    text = ("tempering 0\nhill_prefactor 0.5\nhill_density 200\ndimension 1\nbox_low 0\nbox_high %.17g\n"
            "bias_spacing %.17g\nbias_sigma 0.02\n" % (hi, spacing))
    n = 100_000
    results = {}
    for mode in (-1, 0, 1):                     # the launcher's choice / integrals first / tiles first (forced)
        if mode == 1 and tiles > 700:
            continue                            # (forcing the waiting tiles ahead where they fill the machine is the deadlock the rule avoids)
        cfg = str(workdir / ("order_%d_%d.edm" % (tiles, mode + 1)))
        open(cfg, "w").write(text + "hills_filename %s/H_%d_%d\nhistogram_filename %s/HI_%d_%d\n"
                             % (workdir, tiles, mode + 1, workdir, tiles, mode + 1))
        b = H.Bias(cfg)
        b.setup(1.0, 1.0)
        b.subdivide([0], [hi], [0], [hi], [0], [0.3])
        b.set("debug_tiles_first", mode)
        for step in range(3):
            r = W.pair_distances(n, 700 + step).reshape(-1, 1)
            b.add_hills(r, W.uniform(750 + step, n), -1, est=2 * n)
        v, dv = b.gauss.download()
        results[mode] = (v, dv, b.hist.values, b.get("cum_bias"), b.get("overflow_right"), b.get("hills_added"))
        assert b.get("polled_batches") > 0 and b.get("poll_fallbacks") == 0, "the chained launch completed through its polled word"
        del b
    for mode in results:
        for a, c in zip(results[-1], results[mode]):
            assert np.array_equal(np.asarray(a), np.asarray(c)), "dispatch order %d changed the results" % mode
    # and the hills are the reference's
    cfg = str(workdir / ("order_%d_o.edm" % tiles))
    open(cfg, "w").write(text + "hills_filename %s/H_o%d\nhistogram_filename %s/HI_o%d\n" % (workdir, tiles, workdir, tiles))
    o = B.Bias(oracle_lib, cfg)
    o.setup(1.0, 1.0)
    o.subdivide([0], [hi], [0], [hi], [0], [0.3])
    for step in range(3):
        r = W.pair_distances(n, 700 + step).reshape(-1, 1)
        u = W.uniform(750 + step, n)
        o.pre_add_hill(2 * n)
        for i in np.nonzero(u < 200.0 / (2 * n))[0]:
            o.add_hill(r[i], float(u[i]))
        o.post_add_hill()
    v = results[-1][0]
    assert np.allclose(v, o.gauss.grid.values, rtol=1e-9, atol=1e-13 * np.abs(v).max())
    assert results[-1][4] == o.get("overflow_right") and results[-1][5] == o.get("hills_added")


def test_device_forces_read_from_a_non_blocking_stream_after_wait(oracle_lib, workdir):
    """A forces-only call returns when the energy sums are on the host; the force kernel may not have retired.  A
    consumer on a hipStreamNonBlocking stream of its own (which the handle's blocking stream does not order) calls
    edm_hip_bias_wait() first (include/edm_hip.h, 'Completion'): the array it then copies holds the call's forces."""
    import ctypes as C

    cfg = str(workdir / "wait.edm")
    open(cfg, "w").write("tempering 0\nhill_prefactor 0.5\nhill_density 250\ndimension 1\nbox_low 0\nbox_high 2.8\n"
                         "bias_spacing 0.00025\nbias_sigma 0.025\nhills_filename %s/H\nhistogram_filename %s/HI\n" % (workdir, workdir))
    b = H.Bias(cfg)
    b.setup(1.0, 1.0)
    b.subdivide([0], [2.8], [0], [2.8], [0], [0.3])
    hx = np.zeros((300, 3))
    hx[:, 0] = W.pair_distances(300, 5)
    b.gauss.add_values(hx, 0.01)
    hip = C.CDLL("libamdhip64.so")
    stream = C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(stream), 1) == 0   # hipStreamNonBlocking
    n = 1 << 20
    d_f = H.DeviceArray.zeros((n,))
    out = np.empty(n)
    polled0 = b.get("polled_forces")
    for rep in range(5):
        r = W.pair_distances(n, 40 + rep)
        d_r = H.DeviceArray.from_host(r)
        e = b.pair_forces_device(d_r, d_f, n)
        b.wait()
        assert hip.hipMemcpyAsync(C.c_void_p(out.ctypes.data), C.c_void_p(d_f.ptr), C.c_size_t(out.nbytes), 2, stream) == 0
        assert hip.hipStreamSynchronize(stream) == 0
        assert np.array_equal(out, d_f.to_host())
        assert np.isfinite(e) and np.abs(out).max() > 0
    assert b.get("polled_forces") > polled0, "the calls did return on their polled energy sums"
    hip.hipStreamDestroy(stream)


@pytest.mark.parametrize("words", [8, 1024, 8192, 65536])
def test_completion_flag_never_overtakes_its_data(words):
    """The polled completion publishes its flag with a RELAXED system-scope store behind s_waitcnt(0) + a barrier
    (edm_kernels.hip, limit_and_readback): the same protocol on a region whose every word is the launch's number --
    a host that sees the flag and then finds an older word has caught the flag ahead of its data.  64 B ... 512 KB
    regions (the library's own are <= 64 KB), thousands of launches each, read back last word first."""
    iterations = 20000 if words <= 8192 else 4000
    bad, late = H.flag_order_stress(iterations, words)
    assert late == 0, "a completion flag did not arrive within 200 ms"
    assert bad == 0, "%d words were older than their completion flag" % bad
