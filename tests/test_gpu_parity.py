"""GPU parity tests proper: the HIP path, called through the C ABI (include/edm_hip.h),
against (a) the golden fixtures the real reference produced and (b) the CPU oracle on
seeded inputs, plus size-independent properties at BASELINE.json's full sizes.

Bars (BASELINE.json north_star): node indices, histogram counts and limiter decisions
bit-exact; energies / forces / grid values within 1e-6 relative.  The tolerances asserted
below are much tighter (they are what the kernels achieve: differences come only from
device exp() ulps and fixed-order tree reductions) and are written next to each check.
"""
import hashlib
import os

import numpy as np
import pytest

import edm_amd.hip as H
import edm_amd.workloads as W
from oracle import binding as B

import golden_util as GU

pytestmark = pytest.mark.gpu

RTOL = 1e-9      # asserted relative tolerance for doubles (bar: 1e-6)
ATOL_GRID = 1e-13  # absolute floor for values that cancel to ~0


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    H.require_gpu()  # fail loudly: there is no CPU fallback
    yield


def close(a, b, rtol=RTOL, atol=0.0, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = np.maximum(np.abs(a), np.abs(b))
    bad = np.abs(a - b) > (atol + rtol * scale)
    assert not bad.any(), "%s: %d/%d differ, worst %g (rel %g)" % (
        what, bad.sum(), bad.size, np.abs(a - b).max(), (np.abs(a - b) / np.maximum(scale, 1e-300)).max())


def make_pair(sc, oracle_lib):
    g = H.Gauss.create(sc["lo"], sc["hi"], sc["sp"], sc["per"], 1, sc["sg"])
    o = B.Gauss.create(oracle_lib, sc["lo"], sc["hi"], sc["sp"], sc["per"], 1, sc["sg"])
    if sc.get("bnd"):
        g.set_boundary(*sc["bnd"])
        o.set_boundary(*sc["bnd"])
    return g, o


# ---------------------------------------------------------------------------------
# golden fixtures (outputs of the real reference)
# ---------------------------------------------------------------------------------
@pytest.mark.parametrize("sc", GU.gauss_scenarios(), ids=lambda s: s["name"])
def test_gauss_golden(sc):
    d = GU.gauss_data(sc["name"])
    g = H.Gauss.create(sc["lo"], sc["hi"], sc["sp"], sc["per"], 1, sc["sg"])
    if sc["bnd"]:
        g.set_boundary(*sc["bnd"])
    # geometry: integers exact, host-computed doubles exact
    assert [int(v) for v in g.number] == sc["grid_number"]
    assert g.minisize == sc["minisize"]
    assert [float(v) for v in g.dx] == sc["dx"]
    assert [float(v) for v in g.max] == sc["grid_max"]
    assert [float(v) for v in g.sigma] == sc["sigma_eff"]
    # McGovern-De Pablo tables as they sit in HBM (host libm, gaussian_grid.h:394-431): bit-identical to the reference's
    for dkey, tab in sc["tables"].items():
        t0, t1 = g.bc_tables(int(dkey))
        assert hashlib.sha256(t0.tobytes() + t1.tobytes()).hexdigest() == tab["sha256"], "McGDP tables differ from the reference's"
        assert [float(v) for v in t0[::4099]] == tab["denom"]
    for dd in range(g.dim):
        if str(dd) not in sc["tables"]:
            with pytest.raises(H.EdmHipError):
                g.bc_tables(dd)   # a periodic boundary dimension has no tables
    dim = g.dim
    pad = np.zeros((len(d["hill_x"]), 3))
    pad[:, :dim] = d["hill_x"]
    added = g.add_values(pad, d["hill_h"])
    close(added, d["bias_added"], rtol=1e-10, atol=1e-15, what="bias_added")
    v, dv = g.download()
    vmax = np.abs(d["grid_values"]).max()
    close(v, d["grid_values"], rtol=1e-10, atol=1e-13 * vmax, what="grid values")
    close(dv, d["grid_derivs"], rtol=1e-10, atol=1e-12 * np.abs(d["grid_derivs"]).max(), what="grid derivs")
    q = np.zeros((len(d["queries"]), 3))
    q[:, :dim] = d["queries"]
    E, der = g.get_value_deriv(q)
    close(E, d["E"], rtol=1e-9, atol=1e-13 * vmax, what="E")
    close(der, d["der"], rtol=1e-9, atol=1e-11 * np.abs(d["der"]).max(), what="der")
    flat = g.sample_index(q)
    assert np.array_equal(flat, d["flat_index"]), "node indices must be bit-exact"


def _parse_hills(path):
    rows = []
    for line in open(path):
        t = line.split()
        rows.append((int(t[0]), t[1], int(t[2])) + tuple(float(v) for v in t[3:]))
    return rows


def _grid_file_numbers(path):
    head, nums = [], []
    for line in open(path):
        if line.startswith("#"):
            head.append(line)
        elif line.strip():
            nums.append([float(v) for v in line.split()])
    return head, np.array(nums)


@pytest.mark.parametrize("case", GU.controller_cases(), ids=lambda c: c["name"])
def test_controller_golden(case, workdir):
    name = case["name"]
    hills = str(workdir / ("HILLS_" + name))
    cfg = str(workdir / (name + ".edm"))
    with open(cfg, "w") as fh:
        fh.write(case["cfg"].replace("@FIXTURES@", GU.FIXTURES)
                 + "\nhills_filename %s\nhistogram_filename %s.hist\n" % (hills, hills))
    b = H.Bias(cfg)
    dim = int(b.get("dim"))
    b.setup(1.0, 1.0)
    lo, hi = b.array("min"), b.array("max")
    b.subdivide(lo, hi, lo, hi, case["per"], case["skin"])
    for step in range(case["steps"]):
        pos, ru, mask = GU.controller_inputs(case, step, dim, lo, hi)
        forces = np.zeros_like(pos)
        apply_mask = 1 if step % 2 else -1
        b.set_mask(mask)
        e = b.update_forces(pos, forces, apply_mask)
        close(e, case["E"][step], rtol=1e-9, atol=1e-13, what="energy step %d" % step)
        close(forces[:8, :dim], case["forces_head"][step], rtol=1e-8, atol=1e-11, what="forces")
        if case["mode"] == "explicit":
            b.pre_add_hill(1)
            for k in range(case["n"]):
                b.add_hill(pos[k], float(ru[k]))
            b.post_add_hill()
        else:
            b.add_hills(pos, ru, apply_mask)
        close(b.get("cum_bias"), case["cum"][step], rtol=1e-10, what="cum_bias")
        got = [int(b.get("overflow_left")), int(b.get("overflow_right")), int(b.get("b_skip_hill_add"))]
        assert got == case["overflow"][step], "limiter decisions must match exactly (step %d)" % step
    d = GU.controller_data(name)
    v, dv = b.gauss.download()
    close(v, d["grid_values"], rtol=1e-9, atol=1e-13 * np.abs(d["grid_values"]).max(), what="grid")
    close(dv, d["grid_derivs"], rtol=1e-9, atol=1e-11 * max(np.abs(d["grid_derivs"]).max(), 1e-300), what="derivs")
    assert np.array_equal(b.hist.values, d["hist"]), "histogram counts are integers: exact"
    b.write_bias(str(workdir / "BIAS"))
    b.write_histogram()
    del b
    # HILLS log: same events in the same order; numbers to the printed precision
    got = _parse_hills(hills + "_0")
    want = _parse_hills(os.path.join(GU.GOLDEN, "ctrl_%s.hills.txt" % name))
    assert len(got) == len(want)
    for a, w in zip(got, want):
        assert a[:3] == w[:3], (a, w)
        close(a[3:], w[3:], rtol=0, atol=2e-8, what="HILLS line")
    # histogram file is integer-valued: byte-identical
    assert open(hills + ".hist").read() == open(os.path.join(GU.GOLDEN, "ctrl_%s.hist.grid" % name)).read()
    gold_bias = os.path.join(GU.GOLDEN, "ctrl_%s.bias.grid" % name)
    if os.path.exists(gold_bias):
        h1, n1 = _grid_file_numbers(str(workdir / "BIAS"))
        h2, n2 = _grid_file_numbers(gold_bias)
        assert h1 == h2, "grid file header must be byte-identical"
        close(n1, n2, rtol=0, atol=1.01e-8, what="bias grid file body")
        # (the body is printed with 8 decimals: a last-digit flip needs the device exp's ~1e-16 relative
        #  difference from libm to straddle a rounding boundary -- possible in principle; on every fixture the
        #  reference's tests and generator produced, the files are byte-identical)
        assert open(str(workdir / "BIAS")).read() == open(gold_bias).read(), "written .grid file byte-identical to the reference's"


def test_known_answers(workdir):
    k = GU.kats()
    fx = GU.FIXTURES
    cfg = str(workdir / "nb.edm")
    open(cfg, "w").write(open(os.path.join(fx, "notebook_input.edm")).read() + "\nhills_filename %s/H1\n" % workdir)
    b = H.Bias(cfg)
    b.setup(1, 1)
    b.subdivide([0], [10], [0], [10], [0], [0])
    b.pre_add_hill(1)
    b.add_hill([0.25], 1.0)
    b.post_add_hill()
    E, der = b.gauss.get_value_deriv([[0.24]])
    close([E[0], der[0, 0]], k["notebook"]["published"], rtol=1e-12, what="notebook KAT (EDM.ipynb:103)")
    close(b.get("cum_bias"), k["notebook"]["cum_bias"], rtol=1e-12, what="cum_bias")
    # tabular-bias files (DimmedGrid::multi_write, grid.h:509-674, PLUMED and LAMMPS-table layouts): the
    # re-sampling runs through k_lookup in the reference's operation order -- byte-identical to the reference's files
    mw, lt = str(workdir / "mw.grid"), str(workdir / "lt.ltab")
    b.gauss.multi_write(mw, 0)
    b.gauss.multi_write(lt, 1)
    for got, want in ((mw, "file_notebook_multiwrite.grid"), (lt, "file_notebook_lammps.ltab")):
        a = open(got).read()
        w = open(os.path.join(GU.GOLDEN, want)).read()
        if a != w:
            diff = [(i, x, y) for i, (x, y) in enumerate(zip(a.split("\n"), w.split("\n"))) if x != y]
            raise AssertionError("%s differs from the reference's file in %d lines, first: %r" % (want, len(diff), diff[:3]))
    cfg = str(workdir / "sanity.edm")
    open(cfg, "w").write(open(os.path.join(fx, "sanity.edm")).read() + "\nhills_filename %s/H2\n" % workdir)
    b = H.Bias(cfg)
    b.setup(1, 1)
    b.subdivide([0], [10], [0], [10], [1], [0])
    b.add_hills(np.array([[5.0]]), [1.0])
    s = k["sanity"]
    E, _ = b.gauss.get_value_deriv([[5.0]])
    close(E[0], s["value_at_5"], rtol=1e-12, what="edm_sanity value (edm_test.cpp:886)")
    close(b.get("cum_bias"), s["cum_bias"], rtol=1e-12, what="edm_sanity cum_bias")
    assert b.gauss.size == s["grid_size"]
    _, d = b.gauss.get_value_deriv([[4.99], [5.01]])
    assert d[0, 0] > 0 > d[1, 0]  # forces point away from the hill (edm_test.cpp:889-899)
    b.write_bias("sanity.grid")
    h1, n1 = _grid_file_numbers("sanity.grid")
    h2, n2 = _grid_file_numbers(os.path.join(GU.GOLDEN, "file_sanity_bias.grid"))
    assert h1 == h2
    close(n1, n2, rtol=0, atol=1.01e-8, what="sanity bias file")


# ---------------------------------------------------------------------------------
# seeded inputs against the CPU oracle (sizes the oracle finishes in seconds)
# ---------------------------------------------------------------------------------
ORACLE_CASES = [
    dict(name="c1d", lo=[0.0], hi=[2.8], sp=[0.00025], per=[0], sg=[0.025], nh=300, nq=20000),
    dict(name="c1d_periodic", lo=[0.0], hi=[2.8], sp=[0.0005], per=[1], sg=[0.025], nh=200, nq=20000),
    dict(name="tiny_periodic_stencil_wider_than_grid", lo=[2.0], hi=[10.0], sp=[1.0], per=[1], sg=[1.0], nh=6, nq=200),
    dict(name="c2d_256", lo=[0.0, 0.0], hi=[8.0, 8.0], sp=[1 / 32.0] * 2, per=[1, 1], sg=[0.125] * 2, nh=120, nq=20000),
    dict(name="c2d_mixed", lo=[0.0, 0.0], hi=[8.0, 6.0], sp=[0.05, 0.04], per=[1, 0], sg=[0.2, 0.15], nh=100, nq=10000),
    dict(name="c3d_64", lo=[0.0] * 3, hi=[8.0] * 3, sp=[0.125] * 3, per=[1, 1, 1], sg=[0.25] * 3, nh=60, nq=10000),
    dict(name="c3d_nonperiodic", lo=[0.0] * 3, hi=[4.0] * 3, sp=[0.1, 0.125, 0.2], per=[0, 0, 0], sg=[0.3] * 3, nh=24, nq=5000),
]


@pytest.mark.parametrize("c", ORACLE_CASES, ids=lambda c: c["name"])
def test_vs_oracle_seeded(c, oracle_lib):
    g, o = make_pair(c, oracle_lib)
    dim = g.dim
    lo, hi = np.array(c["lo"]), np.array(c["hi"])
    hx = np.zeros((c["nh"], 3))
    hx[:, :dim] = lo + W.uniform(101, c["nh"] * dim).reshape(-1, dim) * (hi - lo) * 1.1 - 0.05 * (hi - lo)
    hh = 0.05 + W.uniform(102, c["nh"])
    added = g.add_values(hx, hh)
    ref_added = np.array([o.add_value(x[:dim], float(h)) for x, h in zip(hx, hh)])
    close(added, ref_added, rtol=1e-10, atol=1e-15, what="bias_added")
    # hill integrals do not depend on grid contents: the read-only entry point agrees too
    close(g.hill_integrals(hx, hh), ref_added, rtol=1e-10, atol=1e-15, what="hill_integrals")
    v, dv = g.download()
    og = o.grid
    close(v, og.values, rtol=1e-10, atol=1e-13 * np.abs(og.values).max(), what="grid")
    close(dv, og.derivs, rtol=1e-10, atol=1e-12 * np.abs(og.derivs).max(), what="derivs")
    q = np.zeros((c["nq"], 3))
    q[:, :dim] = lo + (W.uniform(103, c["nq"] * dim).reshape(-1, dim) * 1.3 - 0.15) * (hi - lo)
    E, der = g.get_value_deriv(q)
    flat = g.sample_index(q)
    refE = np.zeros(c["nq"])
    refD = np.zeros((c["nq"], dim))
    refI = np.full(c["nq"], -1, dtype=np.int64)
    for i, x in enumerate(q[:, :dim]):
        refE[i], refD[i] = o.get_value_deriv(x)
        xr = x.copy()
        if not o.in_bounds(xr):
            xr = o.remap(xr)
        if o.in_bounds(xr) and og.in_grid(xr):
            refI[i] = og.multi2one(og.get_index(xr))
    assert np.array_equal(flat, refI), "node indices must be bit-exact"
    vmax = np.abs(og.values).max()
    close(E, refE, rtol=1e-9, atol=1e-13 * vmax, what="E")
    close(der, refD, rtol=1e-9, atol=1e-11 * np.abs(refD).max(), what="der")
    # update_forces with a group mask (edm_bias.cpp:287-293)
    mask = (W.splitmix64(104, c["nq"]) % np.uint64(4)).astype(np.int32)
    f = W.uniform(105, c["nq"] * 3).reshape(-1, 3)
    f_ref = f.copy()
    e = g.update_forces(q, f, mask, 2)
    sel = (mask & 2) != 0
    f_ref[sel, :dim] -= refD[sel]
    close(e, refE[sel].sum(), rtol=1e-10, what="masked energy")
    close(f, f_ref, rtol=1e-9, atol=1e-11 * max(1.0, np.abs(refD).max()), what="masked forces")
    assert np.array_equal(f[~sel], f_ref[~sel]), "unmasked rows must not be touched"


def test_stochastic_controller_vs_oracle(oracle_lib, workdir):
    """fix_edm_pair-like hill step: 200k samples, hill_density 250, limit = prefactor."""
    text = ("tempering 0\nhill_prefactor 0.5\nhill_density 250\ndimension 1\nbox_low 0\nbox_high 2.8\n"
            "bias_spacing 0.00025\nbias_sigma 0.025\n")
    cfgs = {}
    for tag in ("gpu", "ora"):
        cfgs[tag] = str(workdir / (tag + ".edm"))
        open(cfgs[tag], "w").write(text + "hills_filename %s/HILLS_%s\nhistogram_filename %s/HIST_%s\n" % (workdir, tag, workdir, tag))
    b = H.Bias(cfgs["gpu"])
    o = B.Bias(oracle_lib, cfgs["ora"])
    for x in (b, o):
        x.setup(1.0, 1.0)
        x.subdivide([0], [2.8], [0], [2.8], [0], [0.3])
    n = 200_000
    for step in range(4):
        r = W.pair_distances(n, 300 + step).reshape(-1, 1)
        u = W.uniform(400 + step, n)
        fg = np.zeros((n, 1))
        fo = np.zeros((n, 1))
        eg = b.update_forces(r, fg)
        eo = o.update_forces(r, fo)
        close(eg, eo, rtol=1e-10, atol=1e-13, what="energy")
        close(fg, fo, rtol=1e-8, atol=1e-11, what="forces")
        b.add_hills(r, u, -1, est=2 * n)
        o.pre_add_hill(2 * n)
        for i in np.nonzero(u < 250.0 / (2 * n))[0]:
            o.add_hill(r[i], float(u[i]))
        o.post_add_hill()
        close(b.get("cum_bias"), o.get("cum_bias"), rtol=1e-10, what="cum_bias")
        assert [b.get(k) for k in ("overflow_left", "overflow_right", "b_skip_hill_add", "hills_added")] == \
               [o.get(k) for k in ("overflow_left", "overflow_right", "b_skip_hill_add", "hills_added")]
    v, dv = b.gauss.download()
    close(v, o.gauss.grid.values, rtol=1e-9, atol=1e-13 * np.abs(v).max(), what="grid")
    assert np.array_equal(b.hist.values, o.hist.values)


# ---------------------------------------------------------------------------------
# BASELINE.json full sizes: size-independent properties + sub-sampled oracle checks
# ---------------------------------------------------------------------------------
def _populate(g, o, dim, lo, hi, nh, seed, h=1e-3):
    hx = np.zeros((nh, 3))
    hx[:, :dim] = np.asarray(lo) + W.uniform(seed, nh * dim).reshape(-1, dim) * (np.asarray(hi) - np.asarray(lo))
    g.add_values(hx, h)
    if o is not None:
        for x in hx:
            o.add_value(x[:dim], h)
    return hx


@pytest.mark.parametrize("npairs", [W.W1_PAIRS, W.W2_PAIRS], ids=["W1_1M_pairs", "W2_38.8M_pairs"])
def test_c1d_full_size(npairs, oracle_lib):
    c = W.C1D
    sc = dict(lo=c["lo"], hi=c["hi"], sp=c["spacing"], per=c["periodic"], sg=c["sigma"])
    g, o = make_pair(sc, oracle_lib)
    hx = _populate(g, o, 1, [0.85], [2.8], 512, 2)
    r = W.pair_distances(npairs, 1 if npairs == W.W1_PAIRS else 11)
    d_r = H.DeviceArray.from_host(r)
    d_f = H.DeviceArray((npairs,))
    e = g.pair_forces_device(d_r, d_f, npairs)
    f = d_f.to_host()
    # property 1: the generic strided update_forces kernel (reference operation order) and the
    # division-free / LDS-window pair kernel agree to rounding (1e-12 relative; bar 1e-6)
    d_f2 = H.DeviceArray.zeros((npairs,))
    e2 = H.C.c_double(0)
    H.check(H.lib().edm_hip_gauss_update_forces(g.h, npairs, d_r.ptr, 1, d_f2.ptr, 1, None, -1, H.C.byref(e2)))
    # (dV/dr is a difference of O(V/dx) terms: ~1e-12 of absolute rounding noise is inherent,
    #  in the reference's own arithmetic too)
    close(d_f2.to_host(), f, rtol=1e-10, atol=1e-11 * np.abs(f).max(), what="pair kernel vs generic kernel")
    close(e, e2.value, rtol=1e-12, what="energy of the two kernels")
    # property 2: the in-kernel energy reduction equals the sum of per-sample energies
    d_E = H.DeviceArray((npairs,))
    H.check(H.lib().edm_hip_gauss_get_value_deriv(g.h, npairs, d_r.ptr, 1, d_E.ptr, None))
    close(e, float(np.sum(d_E.to_host())), rtol=1e-11, what="energy sum")
    # sub-sampled oracle check
    idx = (W.splitmix64(7, 20000) % np.uint64(npairs)).astype(np.int64)
    ref = np.array([o.get_value_deriv([r[i]])[1][0] for i in idx])
    close(f[idx], -ref, rtol=1e-9, atol=1e-12 * np.abs(ref).max(), what="forces vs oracle")
    # property 3: linearity -- adding the same hills with the opposite height cancels
    pad = np.zeros((len(hx), 3))
    pad[:, 0] = hx[:, 0]
    g.add_values(pad, -1e-3)
    v, dv = g.download()
    assert np.abs(v).max() < 1e-15 and np.abs(dv).max() < 1e-13


def test_c2d_full_size(oracle_lib):
    c = W.C2D
    sc = dict(lo=c["lo"], hi=c["hi"], sp=c["spacing"], per=c["periodic"], sg=c["sigma"])
    g, o = make_pair(sc, oracle_lib)
    assert list(g.number) == [2048, 2048]
    _populate(g, o, 2, c["lo"], c["hi"], 250, 22, h=0.01)
    n = 262144
    x = W.atom_positions(n, 21)
    f = np.zeros((n, 3))
    e = g.update_forces(x, f)
    idx = (W.splitmix64(8, 5000) % np.uint64(n)).astype(np.int64)
    ref = np.array([o.get_value_deriv(x[i, :2])[1] for i in idx])
    close(f[idx, :2], -ref, rtol=1e-9, atol=1e-12 * np.abs(ref).max(), what="2-D forces vs oracle")
    assert not f[:, 2].any()
    E, _ = g.get_value_deriv(x)
    close(e, E.sum(), rtol=1e-11, what="2-D energy sum")
    v, _ = g.download()
    close(v, o.grid.values, rtol=1e-10, atol=1e-13 * np.abs(v).max(), what="2-D grid after 250 hills")


def test_c3d_full_size_properties(oracle_lib):
    """512^3 bias grid (4.3 GB of node records): index round trip, linearity, locality."""
    c = W.C3D
    g = H.Gauss.create(c["lo"], c["hi"], c["spacing"], c["periodic"], 1, c["sigma"])
    assert list(g.number) == [512, 512, 512] and g.minisize == [11, 11, 11]
    n = 262144
    x = W.atom_positions(n, 31)
    flat = g.sample_index(x)
    want = (np.floor(x / 0.125).astype(np.int64) * np.array([1, 512, 512 * 512])).sum(axis=1)
    assert np.array_equal(flat, want), "3-D node indices"
    hx = x[:64].copy()
    added = g.add_values(hx, 0.02)
    # a periodic grid's hill integral depends only on the position inside the cell: the oracle on a
    # 64^3 grid of the same spacing/sigma at x mod 8 (= 64 cells exactly) must give the same numbers
    o = B.Gauss.create(oracle_lib, [0.0] * 3, [8.0] * 3, c["spacing"], [1, 1, 1], 1, c["sigma"])
    ref = np.array([o.add_value(np.mod(p, 8.0), 0.02) for p in hx[:16]])
    close(added[:16], ref, rtol=1e-9, what="3-D hill integrals vs oracle on the equivalent 64^3 grid")
    assert np.all(np.abs(added / 0.02 - 1) < 2e-3)  # support cut at dp2 < 8 loses ~0.1% of the mass
    f = np.zeros((n, 3))
    e = g.update_forces(x, f)
    assert e > 0 and np.isfinite(f).all()
    # a sample sitting on a hill centre feels (almost) no force from that hill, and the energy is the peak
    E, der = g.get_value_deriv(hx[:1])
    assert E[0] >= 0.02 / (np.pi ** 1.5 * (0.25 * np.sqrt(2)) ** 3) * 0.99
    # the culled tile-owned gather (ball test per tile) against the oracle: around isolated hills the
    # big grid must carry exactly the one-hill bias of the equivalent 64^3 grid, inside the dp2 < 8
    # ball, across its rim and in the stencil-box corners the tile cull drops
    dist = np.abs(hx[:, None, :] - hx[None, :, :])
    dist = np.sqrt((np.minimum(dist, 64.0 - dist) ** 2).sum(axis=2)) + 1e9 * np.eye(len(hx))
    lonely = [i for i in range(len(hx)) if dist[i].min() > 6.5][:3]
    assert len(lonely) == 3
    for i in lonely:
        o1 = B.Gauss.create(oracle_lib, [0.0] * 3, [8.0] * 3, c["spacing"], [1, 1, 1], 1, c["sigma"])
        o1.add_value(np.mod(hx[i], 8.0), 0.02)
        off = (W.uniform(500 + i, 900).reshape(300, 3) - 0.5) * 2 * 1.2   # the ball has radius 1.0, the stencil box +-1.375
        pts = np.mod(hx[i] + off, 64.0)
        Eg, Dg = g.get_value_deriv(pts)
        ref = [o1.get_value_deriv(np.mod(p, 8.0)) for p in pts]
        Er = np.array([r[0] for r in ref])
        Dr = np.array([r[1] for r in ref])
        assert (Er > 0).sum() > 50 and (Er == 0).sum() > 20
        close(Eg, Er, rtol=1e-9, atol=1e-13 * Er.max(), what="3-D bias around a hill vs oracle")
        close(Dg, Dr, rtol=1e-8, atol=1e-12 * np.abs(Dr).max(), what="3-D bias gradient around a hill vs oracle")
    g.add_values(hx, -0.02)
    E2, _ = g.get_value_deriv(x[:4096])
    assert np.abs(E2).max() < 1e-15, "hills followed by their negatives leave an empty grid"


def test_single_rank_rccl_exchange_matches_plain(workdir):
    """The RCCL hill exchange (all-gather of hill records + all-reduce of the step's bias) with a
    one-rank communicator must reproduce the communicator-free controller bit for bit."""
    text = ("tempering 0\nhill_prefactor 0.5\nhill_density 60\ndimension 1\nbox_low 0\nbox_high 2.8\n"
            "bias_spacing 0.001\nbias_sigma 0.05\n")
    state = []
    for tag in ("plain", "rccl"):
        cfg = str(workdir / (tag + ".edm"))
        open(cfg, "w").write(text + "hills_filename %s/HILLS_%s\nhistogram_filename %s/HIST_%s\n" % (workdir, tag, workdir, tag))
        b = H.Bias(cfg)
        if tag == "rccl":
            b.comm_init(H.comm_unique_id(), 1, 0)
        b.setup(1.0, 1.0)
        b.subdivide([0], [2.8], [0], [2.8], [0], [0.3])
        n = 50_000
        for step in range(5):
            r = W.pair_distances(n, 900 + step).reshape(-1, 1)
            b.add_hills(r, W.uniform(950 + step, n), -1, est=2 * n)
        v, dv = b.gauss.download()
        state.append((v, dv, b.hist.values, b.get("cum_bias"), b.get("overflow_right"), b.get("hills_added")))
        del b
    for a, c in zip(state[0], state[1]):
        assert np.array_equal(np.asarray(a), np.asarray(c))
    assert state[0][0].max() > 0


def test_initial_bias_and_2d_target_vs_oracle(oracle_lib, workdir):
    """initial_bias_filename (Grid::add from a PLUMED file read with interpolation, edm_bias.cpp:166-167)
    and target_filename in 2-D (nearest-lower lookup, :545-546) against the oracle."""
    fx = GU.FIXTURES
    text = ("tempering 0\nhill_prefactor 0.3\nhill_density 25\nbias_per_step 0.2\ndimension 2\nbox_low 0 -3.141593\n"
            "box_high 2.5 3.141593\nbias_spacing 0.04 0.1\nbias_sigma 0.1 0.25\n"
            "target_filename %s/2.grid\ninitial_bias_filename %s/2.grid\n" % (fx, fx))
    cfg = {}
    for tag in ("gpu", "ora"):
        cfg[tag] = str(workdir / (tag + ".edm"))
        open(cfg[tag], "w").write(text + "hills_filename %s/H_%s\nhistogram_filename %s/HIST_%s\n" % (workdir, tag, workdir, tag))
    b = H.Bias(cfg["gpu"])
    o = B.Bias(oracle_lib, cfg["ora"])
    lo, hi = [0, -3.141593], [2.5, 3.141593]
    for x in (b, o):
        x.setup(1.0, 1.0)
        x.subdivide(lo, hi, lo, hi, [0, 1], [0.0, 0.0])
    assert b.get("b_targeting") == 1
    close(b.get("expected_target"), o.get("expected_target"), rtol=1e-14, what="expected_target")
    v, dv = b.gauss.download()
    og = o.gauss.grid
    close(v, og.values, rtol=1e-10, atol=1e-13 * np.abs(og.values).max(), what="grid after initial bias")
    close(dv, og.derivs, rtol=1e-10, atol=1e-12 * np.abs(og.derivs).max(), what="derivs after initial bias")
    n = 3000
    for step in range(3):
        pos = np.zeros((n, 3))
        pos[:, 0] = W.uniform(1200 + step, n) * 2.5
        pos[:, 1] = (W.uniform(1300 + step, n) * 2 - 1) * 3.141593
        u = W.uniform(1400 + step, n)
        b.add_hills(pos, u)
        o.add_hills(np.ascontiguousarray(pos), u)
        close(b.get("cum_bias"), o.get("cum_bias"), rtol=1e-10, what="cum_bias")
        assert [b.get(k) for k in ("overflow_left", "overflow_right", "hills_added")] == \
               [o.get(k) for k in ("overflow_left", "overflow_right", "hills_added")]
    v, dv = b.gauss.download()
    close(v, og.values, rtol=1e-9, atol=1e-12 * np.abs(og.values).max(), what="grid")
    assert np.array_equal(b.hist.values, o.hist.values)


def test_dense_batch_fused_path_vs_oracle(oracle_lib, workdir):
    """Dense hill batches on a small grid take the fused path (gather first, integrals as a by-product,
    limiter, correction gather).  Unlimited batch and a limited controller step that crosses the limit
    inside the batch, both against the oracle's sequential add_value loop."""
    c = dict(lo=[0.0], hi=[2.8], sp=[0.001], per=[0], sg=[0.05])
    g, o = make_pair(c, oracle_lib)
    n = 6000
    hx = np.zeros((n, 3))
    hx[:, 0] = W.pair_distances(n, 1500)
    hh = 1e-4 * (1.0 + W.uniform(1501, n))
    added = g.add_values(hx, hh)
    ref = np.array([o.add_value(x[:1], float(h)) for x, h in zip(hx, hh)])
    close(added, ref, rtol=1e-10, atol=1e-18, what="fused per-hill integrals")
    v, dv = g.download()
    og = o.grid
    close(v, og.values, rtol=1e-9, atol=1e-12 * np.abs(og.values).max(), what="grid (fused)")
    close(dv, og.derivs, rtol=1e-9, atol=1e-11 * np.abs(og.derivs).max(), what="derivs (fused)")
    # limited: all-samples mode, the limit is crossed ~96% into the batch; the rest is deferred
    # (320, 640, 960 buffered hills after the three steps: below the 2048-slot buffer)
    text = ("tempering 0\nhill_prefactor 0.5\nbias_per_step 0.48\ndimension 1\nbox_low 0\nbox_high 2.8\n"
            "bias_spacing 0.001\nbias_sigma 0.05\n")
    cfg = {}
    for tag in ("gpu", "ora"):
        cfg[tag] = str(workdir / (tag + ".edm"))
        open(cfg[tag], "w").write(text + "hills_filename %s/H_%s\nhistogram_filename %s/HIST_%s\n" % (workdir, tag, workdir, tag))
    b = H.Bias(cfg["gpu"])
    ob = B.Bias(oracle_lib, cfg["ora"])
    for x in (b, ob):
        x.setup(1.0, 1.0)
        x.subdivide([0], [2.8], [0], [2.8], [0], [0.3])
    for step in range(3):
        m = 8000
        r = W.pair_distances(m, 1600 + step).reshape(-1, 1)
        u = W.uniform(1700 + step, m)
        b.add_hills(r, u)
        ob.add_hills(np.ascontiguousarray(r), u)
        close(b.get("cum_bias"), ob.get("cum_bias"), rtol=1e-10, what="cum_bias step %d" % step)
        keys = ("overflow_left", "overflow_right", "b_skip_hill_add", "hills_added")
        assert [b.get(k) for k in keys] == [ob.get(k) for k in keys], step
    v, dv = b.gauss.download()
    ogg = ob.gauss.grid
    close(v, ogg.values, rtol=1e-9, atol=1e-12 * np.abs(ogg.values).max(), what="grid (fused + limiter)")
    assert np.array_equal(b.hist.values, ob.hist.values)
    assert ob.get("overflow_right") > 0


@pytest.mark.parametrize("mode", ["fused_launch", "odd_n_other_samples", "count_not_deferred", "all_samples", "communicator"])
def test_pair_step_equals_separate_calls(mode, workdir):
    """edm_hip_bias_pair_step (forces + hill cycle, one host wait) against the separate calls in the
    order fix_edm_pair makes them: pre_add_hill, update_force over the pairs, add_hill per staged
    sample, post_add_hill.  A small bias_per_step keeps the overflow buffer busy, so the flush inside
    pre_add_hill precedes the force evaluation in both.  The modes walk the ways the force kernel is queued:
    in one launch with the step's selection (short pair arrays, deferred hill count), on its own ahead of a
    synchronous selection (expected count above the deferred bound), ahead of an all-samples batch, and in the
    packing launch of the multi-GPU exchange (one-rank communicator)."""
    density = {"count_not_deferred": "hill_density 900\n", "all_samples": ""}.get(mode, "hill_density 40\n")
    per_step = "bias_per_step 0.12\n" if mode != "all_samples" else "bias_per_step 0.004\n"
    text = ("tempering 0\nhill_prefactor 0.5\n" + density + per_step + "dimension 1\nbox_low 0\n"
            "box_high 2.8\nbias_spacing 0.001\nbias_sigma 0.05\n")
    n = 6001 if mode == "odd_n_other_samples" else 6000
    ns = 4000 if mode == "odd_n_other_samples" else (700 if mode == "all_samples" else n)
    state = []
    for tag in ("fused", "separate"):
        cfg = str(workdir / (tag + ".edm"))
        open(cfg, "w").write(text + "hills_filename %s/HILLS_%s\nhistogram_filename %s/HIST_%s\n" % (workdir, tag, workdir, tag))
        b = H.Bias(cfg)
        if mode == "communicator":
            b.comm_init(H.comm_unique_id(), 1, 0)
        b.setup(1.0, 1.0)
        b.subdivide([0], [2.8], [0], [2.8], [0], [0.3])
        energies, forces = [], []
        for step in range(6):
            r = W.pair_distances(n, 300 + step)
            rs = r if ns == n else W.pair_distances(ns, 400 + step)
            u = W.uniform(350 + step, ns)
            d_r = H.DeviceArray.from_host(r)
            d_s = d_r if ns == n else H.DeviceArray.from_host(rs)
            d_u = H.DeviceArray.from_host(u)
            d_f = H.DeviceArray.zeros((n,))
            if tag == "fused":
                e = b.pair_step_device(d_r, d_f, n, d_s, d_u, ns, est=ns)
            else:
                b.pre_add_hill(ns)
                e = b.pair_forces_device(d_r, d_f, n)
                for i in range(ns):
                    b.add_hill([rs[i]], u[i])
                b.post_add_hill()
            energies.append(e)
            forces.append(d_f.to_host())
        v, dv = b.gauss.download()
        state.append((v, dv, b.hist.values, np.array(energies), np.array(forces), b.get("cum_bias"),
                      b.get("overflow_right"), b.get("hills_added")))
        del b
    for a, c in zip(state[0], state[1]):
        assert np.array_equal(np.asarray(a), np.asarray(c))
    assert state[0][0].max() > 0 and state[0][6] > 0   # hills were added and the overflow buffer was used
    assert np.abs(state[0][4][-1]).max() > 0           # the last step's forces saw a non-zero bias
    assert open(str(workdir / "HILLS_fused_0")).read() == open(str(workdir / "HILLS_separate_0")).read()


@pytest.mark.parametrize("dim", [1, 2, 3])
def test_packed_exchange_virtual_ranks(dim, workdir):
    """The multi-GPU packed hill exchange (fixed-size packets, one all-gather, device-side unpack into the
    rank-major list, deferred count) with a one-rank communicator whose packet is replicated three times
    (test hook): must equal, bit for bit, a communicator-free run fed with the samples repeated three
    times -- that IS the rank-major list three identical ranks would produce."""
    if dim == 1:
        text = ("tempering 0\nhill_prefactor 0.5\nhill_density 60\nbias_per_step 0.6\ndimension 1\nbox_low 0\nbox_high 2.8\n"
                "bias_spacing 0.001\nbias_sigma 0.05\n")
        lo, hi, per, skin = [0], [2.8], [0], [0.3]
    elif dim == 2:
        text = ("tempering 0\nhill_prefactor 0.5\nhill_density 60\nbias_per_step 0.6\ndimension 2\nbox_low 0 0\nbox_high 8 8\n"
                "bias_spacing 0.05 0.05\nbias_sigma 0.2 0.2\n")
        lo, hi, per, skin = [0, 0], [8, 8], [1, 1], [0, 0]
    else:
        text = ("tempering 0\nhill_prefactor 0.5\nhill_density 60\nbias_per_step 0.6\ndimension 3\nbox_low 0 0 0\nbox_high 6 6 6\n"
                "bias_spacing 0.1 0.1 0.1\nbias_sigma 0.25 0.25 0.25\n")
        lo, hi, per, skin = [0] * 3, [6] * 3, [1, 0, 1], [0, 0.3, 0]
    n, R = 40_000, 3
    state = []
    for tag in ("plain", "virtual"):
        cfg = str(workdir / (tag + ".edm"))
        open(cfg, "w").write(text + "hills_filename %s/HILLS_%s\nhistogram_filename %s/HIST_%s\n" % (workdir, tag, workdir, tag))
        b = H.Bias(cfg)
        if tag == "virtual":
            b.comm_init(H.comm_unique_id(), 1, 0)
            b.set("debug_virtual_ranks", R)
        b.setup(1.0, 1.0)
        b.subdivide(lo, hi, lo, hi, per, skin)
        for step in range(5):
            r = (W.pair_distances(n, 700 + step).reshape(-1, 1) if dim == 1 else
                 W.uniform(700 + step, n * dim).reshape(n, dim) * np.array(hi))
            u = W.uniform(750 + step, n)
            if tag == "plain":
                b.add_hills(np.tile(r, (R, 1)), np.tile(u, R), -1, est=2 * n)
            else:
                b.add_hills(r, u, -1, est=2 * n)
        v, dv = b.gauss.download()
        state.append((v, dv, b.hist.values, b.get("cum_bias"), b.get("overflow_right"), b.get("hills_added")))
        del b
    for a, c in zip(state[0], state[1]):
        assert np.array_equal(np.asarray(a), np.asarray(c))
    assert state[0][0].max() > 0 and state[0][5] > 30
    assert open(str(workdir / "HILLS_plain_0")).read() == open(str(workdir / "HILLS_virtual_0")).read()


def test_python_package_mirror_notebook(workdir):
    """The reference's Python face (python/edm: EDMBias(input, T, kB), set_box, add_hill, get_force) on the
    GPU path, driven exactly like the published notebook (python-example/EDM.ipynb): one hill at 0.25,
    get_force(0.24) must return the notebook's printed numbers."""
    import edm_amd.edm as E

    k = GU.kats()
    cfg = str(workdir / "nb2.edm")
    open(cfg, "w").write(open(os.path.join(GU.FIXTURES, "notebook_input.edm")).read() + "\nhills_filename %s/H9\n" % workdir)
    bias = E.EDMBias(cfg, 1.0, 1.0)
    bias.set_box([0], [10], [0])
    bias.add_hill([0.25])
    e, f = bias.get_force([0.24])
    close([e, f[0]], k["notebook"]["published"], rtol=1e-12, what="notebook KAT through the python package mirror")
    close(bias.cum_bias, k["notebook"]["cum_bias"], rtol=1e-12, what="cum_bias")
    bias.write_bias(str(workdir / "nb2.bias"))
    bias.write_histogram()
    bias.clear_histogram()
    assert os.path.getsize(str(workdir / "nb2.bias")) > 1000


@pytest.mark.parametrize("mode,n,limit", [("virtual3", 12000, 39.0), ("virtual3", 4500, 24.0), ("rccl1", 12000, 39.0)])
def test_sharded_dense_application(mode, n, limit, workdir):
    steps = 3 if n > 5000 else 1   # (the low limit fills most of the 2048-record overflow buffer in one step)
    """Dense (all-samples) hill batches on the replicated 1-D grid, applied shard-wise: every rank gathers
    only its own slice of the global list into a delta grid, integrals and delta grids are summed over the
    ranks, the limiter runs on the global ordered list.  `virtual3`: three ranks' slices processed one
    after the other in this process (test hook, exercises the slice offsets and the limiter tail that
    starts inside a slice -- or, with the low limit, before the last slice); `rccl1`: a one-rank RCCL communicator (the real ncclAllReduce calls).
    Both must agree with the unsharded controller to rounding: the sums are associated differently."""
    text = ("tempering 0\nhill_prefactor 40\nbias_per_step %g\ndimension 1\nbox_low 0\nbox_high 2.8\n"
            "bias_spacing 0.001\nbias_sigma 0.05\n" % limit)
    state = []
    for tag in ("plain", mode):
        cfg = str(workdir / (tag + ".edm"))
        open(cfg, "w").write(text + "hills_filename %s/HILLS_%s\nhistogram_filename %s/HIST_%s\n" % (workdir, tag, workdir, tag))
        b = H.Bias(cfg)
        if tag == "rccl1":
            b.comm_init(H.comm_unique_id(), 1, 0)
        if tag == "virtual3":
            b.set("debug_virtual_ranks", 3)
        b.setup(1.0, 1.0)
        b.subdivide([0], [2.8], [0], [2.8], [0], [0.3])
        b.set_hill_log(False)
        for step in range(steps):
            r = W.pair_distances(n, 40 + step).reshape(-1, 1)
            b.add_hills(r, W.uniform(45 + step, n), -1, est=n)
        v, dv = b.gauss.download()
        state.append((v, dv, b.hist.values, b.get("cum_bias"), b.get("overflow_left"), b.get("overflow_right"),
                      b.get("hills_added"), b.get("b_skip_hill_add")))
        del b
    p, q = state
    close(q[0], p[0], rtol=1e-12, atol=1e-14 * np.abs(p[0]).max(), what="grid values, sharded vs plain")
    close(q[1], p[1], rtol=1e-11, atol=1e-12 * np.abs(p[1]).max(), what="grid derivatives, sharded vs plain")
    assert np.array_equal(p[2], q[2]), "histogram"
    close(q[3], p[3], rtol=1e-12, what="cum_bias")
    assert p[4:] == q[4:], "limiter decisions (overflow buffer indices, hills added, skip flag)"
    assert p[5] > 0 and p[0].max() > 0, "the limiter must have been active"


def test_step_equals_separate_calls(workdir):
    """edm_hip_bias_step (update_forces + add_hills, one host wait) against the two separate calls, 2-D
    coordinate CV with a group mask and an active limiter: bit-identical forces, energies, grid, histogram."""
    text = ("tempering 0\nhill_prefactor 0.4\nhill_density 30\nbias_per_step 0.25\ndimension 2\nbox_low 0 0\n"
            "box_high 8 8\nbias_spacing 0.05 0.05\nbias_sigma 0.2 0.2\n")
    n = 5000
    state = []
    for tag in ("fused", "separate", "host_arrays"):
        cfg = str(workdir / (tag + ".edm"))
        open(cfg, "w").write(text + "hills_filename %s/HILLS_%s\nhistogram_filename %s/HIST_%s\n" % (workdir, tag, workdir, tag))
        b = H.Bias(cfg)
        b.setup(1.0, 1.0)
        b.subdivide([0, 0], [8, 8], [0, 0], [8, 8], [1, 1], [0.3, 0.3])
        energies, forces = [], []
        for step in range(5):
            x = W.uniform(800 + step, 3 * n).reshape(n, 3) * 8.0
            u = W.uniform(850 + step, n)
            mask = (W.splitmix64(870 + step, n) % np.uint64(4)).astype(np.int32)
            d_x = H.DeviceArray.from_host(np.ascontiguousarray(x))
            d_u = H.DeviceArray.from_host(u)
            d_f = H.DeviceArray.zeros((n, 3))
            b.set_mask(mask)
            if tag == "host_arrays":
                # edm_hip_bias_step_host: positions up, the bias-force DELTA down and added to the host force array,
                # which is never uploaded -- started from zero it ends with the same doubles as the device array
                fh = np.zeros((n, 3))
                e = b.step_host(np.ascontiguousarray(x), fh, mask=mask, runiform=u, apply_mask=1, hill_step=True)
                f0 = W.uniform(990 + step, 3 * n).reshape(n, 3) - 0.5
                f1 = f0.copy()
                e2 = b.step_host(np.ascontiguousarray(x), f1, mask=mask, apply_mask=1, hill_step=False)   # forces only
                d_chk = H.DeviceArray.zeros((n, 3))
                ec = H.C.c_double(0)
                b.set_mask(mask)
                H.check(H.lib().edm_hip_bias_update_forces(b.h, n, d_x.ptr, 3, d_chk.ptr, 3, 1, H.C.byref(ec)))
                assert e2 == ec.value and np.array_equal(f1, f0 + d_chk.to_host()), "delta added to a non-zero force array"
                energies.append(e)
                forces.append(fh)
                continue
            if tag == "fused":
                e = b.step_device(d_x, 3, d_f, 3, n, d_u, apply_mask=1)
            else:
                e = H.C.c_double(0)
                H.check(H.lib().edm_hip_bias_update_forces(b.h, n, d_x.ptr, 3, d_f.ptr, 3, 1, H.C.byref(e)))
                e = e.value
                b.add_hills_device(d_x, n, 3, d_u, 1, -1)
            energies.append(e)
            forces.append(d_f.to_host())
        v, dv = b.gauss.download()
        state.append((v, dv, b.hist.values, np.array(energies), np.array(forces), b.get("cum_bias"),
                      b.get("overflow_right"), b.get("hills_added")))
        del b
    for other in (1, 2):
        for a, c in zip(state[0], state[other]):
            assert np.array_equal(np.asarray(a), np.asarray(c))
    assert state[0][0].max() > 0 and state[0][7] > 10 and np.abs(state[0][4]).max() > 0


@pytest.mark.parametrize("layout", ["lammps_rows", "wide_rows_and_mask"])
@pytest.mark.parametrize("dim", [2, 3])
def test_step_host_helper_threads_and_pieces(dim, layout, workdir):
    """edm_hip_bias_step_host on enough atoms that the delta comes down in several pieces and helper threads share the
    add (odd atom count: the copy kernel's last double, the pieces' rounding): bit-identical to the device-array step,
    with one adding thread and with five, on a force array that starts non-zero.  LAMMPS' own layout (rows of three)
    and rows wider than that with a group mask (position rows of five doubles, force rows of four)."""
    n = 300001
    span = 16.0
    xs, fs, masked = (3, 3, False) if layout == "lammps_rows" else (5, 4, True)
    text = ("tempering 0\nhill_prefactor 0.4\nhill_density 40\nbias_per_step 0.3\ndimension %d\nbox_low %s\n"
            "box_high %s\nbias_spacing %s\nbias_sigma %s\n" % (
                dim, " ".join(["0"] * dim), " ".join(["16"] * dim), " ".join(["0.125"] * dim), " ".join(["0.4"] * dim)))
    results = []
    for tag, threads in (("device", 0), ("host1", 1), ("host5", 5)):
        cfg = str(workdir / ("sh_%s_%d_%s.edm" % (tag, dim, layout)))
        open(cfg, "w").write(text + "hills_filename %s/HILLS_sh_%s%d%s\nhistogram_filename %s/HIST_sh_%s%d%s\n" % (
            workdir, tag, dim, layout, workdir, tag, dim, layout))
        b = H.Bias(cfg)
        b.setup(1.0, 1.0)
        b.subdivide([0] * dim, [span] * dim, [0] * dim, [span] * dim, [1] * dim, [0.0] * dim)
        if threads:
            b.set("host_add_threads", threads)
            assert b.get("host_add_threads") == threads
        out = []
        for step in range(3):
            x = np.ascontiguousarray(W.uniform(4100 + step, xs * n).reshape(n, xs) * span)
            u = W.uniform(4200 + step, n)
            f0 = W.uniform(4300 + step, fs * n).reshape(n, fs) - 0.5
            mask = (W.splitmix64(4400 + step, n) % np.uint64(4)).astype(np.int32) if masked else None
            if threads:
                f = f0.copy()
                e = b.step_host(x, f, mask=mask, runiform=u, apply_mask=1 if masked else -1, hill_step=True, est=n)
                out.append((e, f))
            else:
                d_x, d_u = H.DeviceArray.from_host(x), H.DeviceArray.from_host(u)
                d_f = H.DeviceArray.zeros((n, fs))
                if masked:
                    b.set_mask(mask)
                e = b.step_device(d_x, xs, d_f, fs, n, d_u, 1 if masked else -1, n)
                out.append((e, f0 + d_f.to_host()))
        results.append((out, b.gauss.download(), b.get("hills_added"), b.get("cum_bias")))
        del b
    for other in results[1:]:
        for (e0, f0), (e1, f1) in zip(results[0][0], other[0]):
            assert e0 == e1 and np.array_equal(f0, f1)
        assert np.array_equal(results[0][1][0], other[1][0]) and results[0][2:] == other[2:]
    assert results[0][2] > 10
    if dim < fs:   # columns beyond the dimension are the caller's own, untouched
        assert np.array_equal(results[1][0][-1][1][:, dim:], (W.uniform(4302, fs * n).reshape(n, fs) - 0.5)[:, dim:])


def test_large_selection_vs_oracle(oracle_lib, workdir):
    """Stochastic selection over 8 M samples (3907 selection workgroups: the multi-pass branch of the chained
    scan) against the oracle: same accepted hills in the same order -> same grid, histogram, counters."""
    text = ("tempering 0\nhill_prefactor 0.5\nhill_density 300\nbias_per_step 0.4\ndimension 1\nbox_low 0\nbox_high 2.8\n"
            "bias_spacing 0.001\nbias_sigma 0.05\n")
    cfgs = {}
    for tag in ("gpu", "ora"):
        cfgs[tag] = str(workdir / (tag + ".edm"))
        open(cfgs[tag], "w").write(text + "hills_filename %s/HILLS_%s\nhistogram_filename %s/HIST_%s\n" % (workdir, tag, workdir, tag))
    b = H.Bias(cfgs["gpu"])
    o = B.Bias(oracle_lib, cfgs["ora"])
    for x in (b, o):
        x.setup(1.0, 1.0)
        x.subdivide([0], [2.8], [0], [2.8], [0], [0.3])
    n = 8_000_000
    for step in range(2):
        r = W.pair_distances(n, 60 + step)
        u = W.uniform(65 + step, n)
        pos = np.zeros((n, 3))
        pos[:, 0] = r
        o.add_hills(pos, u, -1)
        del pos
        d_r = H.DeviceArray.from_host(r)
        d_u = H.DeviceArray.from_host(u)
        b.add_hills_device(d_r, n, 1, d_u, -1, n)
        close(b.get("cum_bias"), o.get("cum_bias"), rtol=1e-10, what="cum_bias")
        for k in ("overflow_left", "overflow_right", "b_skip_hill_add", "hills_added"):
            assert int(b.get(k)) == int(o.get(k)), k
    v, dv = b.gauss.download()
    og = o.gauss.grid
    close(v, og.values, rtol=1e-9, atol=1e-13 * np.abs(og.values).max(), what="grid")
    assert np.array_equal(b.hist.values, o.hist.values)
    assert int(b.get("hills_added")) > 150


def test_device_rng_equals_explicit_uniforms(workdir):
    """Fast mode of the random numbers (edm_hip_bias_set_device_rng): the uniforms drawn on the device for
    add_hill cycle c are output i + 1 of SplitMix64(seed + c * 0x632BE59BD9B4E019) -- feeding exactly those
    numbers through the explicit-uniform path must give the same bias, bit for bit (both the chained
    selection of large steps and the synchronous selection of small ones)."""
    text = ("tempering 0\nhill_prefactor 0.5\nhill_density 80\nbias_per_step 0.3\ndimension 1\nbox_low 0\nbox_high 2.8\n"
            "bias_spacing 0.001\nbias_sigma 0.05\n")
    seed, K, M = 20261004, 0x632BE59BD9B4E019, (1 << 64) - 1
    sizes = [50_000, 300, 50_000, 2_000]
    state = []
    for tag in ("device", "explicit"):
        cfg = str(workdir / (tag + ".edm"))
        open(cfg, "w").write(text + "hills_filename %s/HILLS_%s\nhistogram_filename %s/HIST_%s\n" % (workdir, tag, workdir, tag))
        b = H.Bias(cfg)
        b.setup(1.0, 1.0)
        b.subdivide([0], [2.8], [0], [2.8], [0], [0.3])
        if tag == "device":
            b.set_device_rng(True, seed)
        for cycle, n in enumerate(sizes):
            d_r = H.DeviceArray.from_host(W.pair_distances(n, 900 + cycle))
            if tag == "device":
                b.add_hills_device(d_r, n, 1, None, -1, n)
            else:
                d_u = H.DeviceArray.from_host(W.uniform((seed + cycle * K) & M, n))
                b.add_hills_device(d_r, n, 1, d_u, -1, n)
        v, dv = b.gauss.download()
        state.append((v, dv, b.hist.values, b.get("cum_bias"), b.get("overflow_right"), b.get("hills_added")))
        del b
    for a, c in zip(state[0], state[1]):
        assert np.array_equal(np.asarray(a), np.asarray(c))
    assert state[0][0].max() > 0
    assert open(str(workdir / "HILLS_device_0")).read() == open(str(workdir / "HILLS_explicit_0")).read()


@pytest.mark.parametrize("reference_order", [False, True], ids=["batch_order", "reference_order"])
def test_pair_list_step_vs_oracle(oracle_lib, workdir, reference_order):
    """(reference_order: edm_hip_bias_set "reference_order" 1 -- the oracle then deposits an entry's hills right behind
    its update_force, lammps/fix_edm_pair.cpp:215-237, so later entries of the step already feel them.)
    fix edm_pair on a device-resident neighbour list (edm_hip_bias_pair_list_step): positions + flattened half
    list in, pair distances / lookups / pair forces / hills on the GPU.  Against the oracle executing the
    reference's per-pair loop with the same uniforms (device stream, sample index 2 * entry + slot): energies,
    forces on owned and ghost atoms, add_hill call counts, limiter decisions, final grid and histogram.  Two atom
    types with a type filter, ghost atoms (j >= nlocal: no force on j, one hill instead of two)."""
    text = ("tempering 0\nhill_prefactor 0.3\nhill_density 25\nbias_per_step 0.2\ndimension 1\nbox_low 0\nbox_high 2.8\n"
            "bias_spacing 0.001\nbias_sigma 0.05\n")
    cfgs = {}
    for tag in ("gpu", "ora"):
        cfgs[tag] = str(workdir / (tag + ".edm"))
        open(cfgs[tag], "w").write(text + "hills_filename %s/HILLS_%s\nhistogram_filename %s/HIST_%s\n" % (workdir, tag, workdir, tag))
    b = H.Bias(cfgs["gpu"])
    o = B.Bias(oracle_lib, cfgs["ora"])
    for x in (b, o):
        x.setup(1.0, 1.0)
        x.subdivide([0], [2.8], [0], [2.8], [0], [0.3])
    seed, K, M = 4242, 0x632BE59BD9B4E019, (1 << 64) - 1
    b.set_device_rng(True, seed)
    b.set("reference_order", 1 if reference_order else 0)
    rng = np.random.default_rng(3)
    nall, nlocal = 700, 520
    types = rng.integers(1, 3, nall).astype(np.int32)          # types 1 and 2; the fix pairs type 1 with type 2
    itype, jtype = 1, 2
    est = nall
    cycle = 0
    for step in range(5):
        x = rng.uniform(0, 9.0, (nall, 3))
        # half list in "neighbour-list order": for owned i, its neighbours j > i within the cutoff (owned or ghost)
        pi, pj = [], []
        for i in range(nlocal):
            d2 = ((x[i + 1:] - x[i]) ** 2).sum(axis=1)
            for j in (np.nonzero(d2 < 2.8 * 2.8)[0] + i + 1):
                pi.append(i)
                pj.append(int(j))
        P = len(pi)
        pi_a, pj_a = np.array(pi, dtype=np.int32), np.array(pj, dtype=np.int32)
        d_fd = H.DeviceArray.zeros((nall, 3))
        hill = step % 2 == 0
        b.pair_list_upload(pi_a, pj_a, types)
        e, ncalls = b.pair_list_step_device(nlocal, itype, jtype, H.DeviceArray.from_host(x), d_fd, hill, est)
        fd = d_fd.to_host()
        # ---- oracle: the reference's loop ----
        if hill:
            o.pre_add_hill(est)
        E, fref, calls = 0.0, np.zeros((nall, 3)), 0
        u = W.uniform((seed + cycle * K) & M, 2 * P)
        staged = []
        for p, (i, j) in enumerate(zip(pi, pj)):
            ti, tj = types[i], types[j]
            if ti == itype:
                ok = tj == jtype
            elif ti == jtype:
                ok = tj == itype
            else:
                ok = False
            if not ok:
                continue
            dvec = x[i] - x[j]
            r = np.sqrt((dvec ** 2).sum())
            dvec = dvec * (1.0 / r)
            ev, fv = o.update_force([r])
            E += ev
            fref[i] += dvec * fv[0]
            if j < nlocal:
                fref[j] -= dvec * fv[0]
            if hill:
                staged.append((r, u[2 * p]))
                calls += 1
                if j < nlocal:
                    staged.append((r, u[2 * p + 1]))
                    calls += 1
                if reference_order:
                    for rr, uu in staged:
                        o.add_hill([rr], float(uu))
                    staged = []
        if hill:
            for r, uu in staged:
                o.add_hill([r], float(uu))
            o.post_add_hill()
            cycle += 1
            assert ncalls == calls, (step, ncalls, calls)
            est = calls
            close(b.get("cum_bias"), o.get("cum_bias"), rtol=1e-9, what="cum_bias")
            for k in ("overflow_left", "overflow_right", "b_skip_hill_add", "hills_added"):
                assert int(b.get(k)) == int(o.get(k)), (step, k)
        close(e, E, rtol=1e-9, atol=1e-13, what="energy step %d" % step)
        close(fd, fref, rtol=1e-8, atol=1e-11 * max(np.abs(fref).max(), 1e-300), what="forces step %d" % step)
        assert not fd[nlocal:].any(), "newton off: ghost atoms receive no force"
    v, _ = b.gauss.download()
    og = o.gauss.grid
    close(v, og.values, rtol=1e-9, atol=1e-13 * np.abs(og.values).max(), what="grid")
    assert np.array_equal(b.hist.values, o.hist.values)
    assert og.values.max() > 0


def test_polled_completion_equals_stream_wait(workdir):
    """Short hill batches hand their results to the host through host-mapped memory flagged by the limiter's
    workgroup, and the call returns while the gather still runs.  Everything a caller can
    observe straight behind such a call -- energies, forces, lookups, the histogram, written files, the final
    grid -- must be identical, byte for byte, to a run with EDM_HIP_POLL=0 (every batch waits for its stream)."""
    import subprocess
    import sys

    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "poll_worker.py")
    digests = []
    releases = []
    polled_forces = []
    # polled (short unlogged batches the limiter leaves alone are released by their header line), polled without the
    # header-line release, and every batch waiting for its stream
    # ... and with every gather tile taking the boundary-duplication ticket instead of the tiles near the walls only
    # ... and with the fix edm step's force kernel in a launch of its own instead of the flush preparation's
    lookup_prep = []
    for tag, poll, header, dup_all in (("polled", None, None, None), ("polled_no_header", None, "0", None), ("waited", "0", None, None),
                                       ("dup_ticket_all", None, None, "1"), ("lookup_alone", None, None, None),
                                       ("integrals_ticket", None, None, None)):
        d = workdir / tag
        d.mkdir()
        env = dict(os.environ)
        env.pop("EDM_HIP_POLL", None)
        env.pop("EDM_HIP_TEST_FORCE", None)
        force = []   # EDM_HIP_TEST_FORCE tokens (csrc/edm_gauss.cpp:test_force): the paths production takes under other conditions
        if tag == "lookup_alone":
            force.append("no_lookup_prep")
        if tag == "integrals_ticket":   # (the per-hill integrals behind a last-arrival ticket, not as tagged stores)
            force.append("no_tagged_integrals")
        if poll is not None:
            env["EDM_HIP_POLL"] = poll
        if header is not None:
            force.append("no_fast_header")
        if dup_all is not None:
            force.append("dup_ticket_all")
        if force:
            env["EDM_HIP_TEST_FORCE"] = ",".join(force)
        res = subprocess.run([sys.executable, worker, str(d)], env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
        lines = [ln for ln in res.stdout.splitlines() if ln.startswith("DIGEST ")]
        assert lines, res.stdout[-2000:]
        digests.append(lines[-1])
        releases.append(int([ln for ln in res.stdout.splitlines() if ln.startswith("HEADER_RELEASES ")][-1].split()[1]))
        polled_forces.append(int([ln for ln in res.stdout.splitlines() if ln.startswith("POLLED_FORCES ")][-1].split()[1]))
        lookup_prep.append(int([ln for ln in res.stdout.splitlines() if ln.startswith("LOOKUP_PREP ")][-1].split()[1]))
    assert digests[0] == digests[1] == digests[2] == digests[3] == digests[4] == digests[5]
    assert lookup_prep[0] >= 8 and lookup_prep[4] == 0, lookup_prep
    assert releases[0] >= 10 and releases[1] == 0 and releases[2] == 0, releases
    assert polled_forces[0] >= 9 and polled_forces[2] == 0, polled_forces   # (forces-only calls: tagged sums / stream wait)


def test_polled_completion_is_what_releases_short_batches(workdir):
    """Telemetry of the polled completion: with the default environment every short stochastic step is released by
    the polled word, none by the stream-wait fallback (a fallback would be correct but slow, and silent)."""
    if os.environ.get("EDM_HIP_POLL", "1")[:1] == "0":
        pytest.skip("polling disabled in the environment")
    cfg = str(workdir / "tele.edm")
    open(cfg, "w").write("tempering 0\nhill_prefactor 0.5\nhill_density 40\nbias_per_step 0.12\ndimension 1\nbox_low 0\n"
                         "box_high 2.8\nbias_spacing 0.001\nbias_sigma 0.05\nhills_filename %s/HILLS_t\n"
                         "histogram_filename %s/HIST_t\n" % (workdir, workdir))
    b = H.Bias(cfg)
    b.setup(1.0, 1.0)
    b.subdivide([0], [2.8], [0], [2.8], [0], [0.3])
    n = 6000
    for step in range(20):
        d_r = H.DeviceArray.from_host(W.pair_distances(n, 700 + step))
        d_u = H.DeviceArray.from_host(W.uniform(750 + step, n))
        d_f = H.DeviceArray.zeros((n,))
        b.pair_step_device(d_r, d_f, n, d_r, d_u, n, est=n)
    # (a fallback -- the 2 ms poll budget ran out and the stream wait took over -- is legitimate under a profiler or a
    #  busy GPU; what must hold is that polling is what normally releases the host)
    assert b.get("poll_fallbacks") <= max(2, b.get("polled_batches") // 10)
    assert b.get("polled_batches") >= 20


@pytest.mark.parametrize("pinned", [False, True], ids=["pageable", "pinned"])
def test_pair_step_host_equals_device_arrays(pinned, workdir):
    """edm_hip_bias_pair_step_host (host arrays staged by the library: distances up, force kernel, forces down while the
    samples go up, hill cycle) against edm_hip_bias_pair_step on arrays resident in HBM: energies, forces, grid,
    histogram and limiter state bit for bit; separate sample arrays and samples aliasing the distances; explicit
    uniforms and the device stream."""
    text = ("tempering 0\nhill_prefactor 0.5\nhill_density 40\nbias_per_step 0.12\ndimension 1\nbox_low 0\n"
            "box_high 2.8\nbias_spacing 0.001\nbias_sigma 0.05\n")
    n, ns = 50000, 70000
    state = []
    for tag in ("host", "device"):
        cfg = str(workdir / (tag + ".edm"))
        open(cfg, "w").write(text + "hills_filename %s/HILLS_%s\nhistogram_filename %s/HIST_%s\n" % (workdir, tag, workdir, tag))
        b = H.Bias(cfg)
        b.setup(1.0, 1.0)
        b.subdivide([0], [2.8], [0], [2.8], [0], [0.3])
        alloc = (lambda m: H.pinned_array(m)) if pinned else (lambda m: np.empty(m))
        h_r, h_f, h_s, h_u = alloc(n), alloc(n), alloc(ns), alloc(ns)
        out = []
        for step in range(6):
            h_r[:] = W.pair_distances(n, 300 + step)
            h_s[:] = W.pair_distances(ns, 400 + step)
            h_u[:] = W.uniform(350 + step, ns)
            alias = step % 3 == 2        # the staged samples ARE the pair distances
            rng = step == 4              # uniforms drawn on the device
            if rng:
                b.set_device_rng(True, 99)
            samples, nsamp = (h_r, n) if alias else (h_s, ns)
            uniforms = None if rng else h_u[:nsamp]
            if tag == "host":
                h_f[:] = -7.0
                e = b.pair_step_host(h_r, h_f, samples[:nsamp], uniforms, est=nsamp)
                f = h_f.copy()
            else:
                d_r = H.DeviceArray.from_host(h_r)
                d_s = d_r if alias else H.DeviceArray.from_host(h_s)
                d_u = None if rng else H.DeviceArray.from_host(np.ascontiguousarray(h_u[:nsamp]))
                d_f = H.DeviceArray.zeros((n,))
                e = b.pair_step_device(d_r, d_f, n, d_s, d_u, nsamp, est=nsamp)
                f = d_f.to_host()
            if rng:
                b.set_device_rng(False, 0)
            out.append((e, f, [b.get(k) for k in ("cum_bias", "overflow_left", "overflow_right", "b_skip_hill_add", "hills_added")]))
        v, dv = b.gauss.download()
        state.append((out, v, dv, b.hist.values))
        del b
    for (e1, f1, s1), (e2, f2, s2) in zip(state[0][0], state[1][0]):
        # (the force kernel runs on its own in the host entry and inside the selection's launch in the other: the
        #  per-pair forces are the same bits, the energy is a sum over differently shaped workgroups)
        assert np.array_equal(f1, f2) and s1 == s2 and abs(e1 - e2) <= 1e-12 * abs(e2)
    for k in (1, 2, 3):
        assert np.array_equal(state[0][k], state[1][k])
    assert state[0][1].max() > 0 and state[0][0][-1][2][2] > 0


def test_ball_list_equals_box_walk():
    """2-D / 3-D per-hill integrals walk the host's list of the stencil offsets that can lie inside the support
    (Tables::ball) instead of the reference's (2 minisize + 1)^D box (gaussian_grid.h:227-281): the same terms in another
    order, so the two walks agree to rounding -- periodic and walled boundaries, hills on walls, seams and nodes."""
    import json
    import subprocess
    import sys

    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ball_worker.py")
    res = {}
    for tag, token in (("list", None), ("box", "no_ball_list"), ("unchained", "no_add_values_chain")):
        env = dict(os.environ)
        env.pop("EDM_HIP_TEST_FORCE", None)
        if token is not None:
            env["EDM_HIP_TEST_FORCE"] = token
        r = subprocess.run([sys.executable, worker], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("INTEGRALS ")][-1]
        raw = json.loads(line[len("INTEGRALS "):])
        res[tag] = {k: (vs if k.endswith("_grid") else np.array([float.fromhex(v) for v in vs])) for k, vs in raw.items()}
    for name in res["list"]:
        if name.endswith("_grid"):
            continue
        a, b = res["list"][name], res["box"][name]
        assert a.shape == b.shape and np.abs(b).max() > 0, name
        assert np.allclose(a, b, rtol=1e-12, atol=1e-14 * np.abs(b).max()), (name, np.abs(a - b).max())
    # a short add_values batch through the chained limiter launch (limit nothing reaches) against the launches of the
    # unlimited path: the same per-hill bias and the same grid bit for bit, the total to rounding (another summation order)
    for name in res["list"]:
        a, u = res["list"][name], res["unchained"][name]
        if name.endswith("_grid"):
            assert a == u, name
        elif name.endswith("_addv_total"):
            assert abs(a[0] - u[0]) <= 1e-13 * abs(u[0]), name
        else:
            assert np.array_equal(a, u), name
