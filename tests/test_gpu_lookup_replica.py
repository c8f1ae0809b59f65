"""The lookup replica of 2-D / 3-D coordinate-CV grids (edm_hip_gauss_set_lookup_replica): lookups through it must
be BIT-IDENTICAL to lookups on the node records -- the same arithmetic spread over four lanes per sample -- after every way the
grid can be written: short hill batches (the in-place tile-owned gather maintains the replica node by node; no
rebuild may happen), dense batches / uploads / clears / Grid::add (replica marked stale, rebuilt on the next lookup),
through the gaussian-grid entry points, the controller's fused step, and at BASELINE's full sizes (2048^2, 512^3)
where the automatic mode switches it on.  Reference arithmetic: lib/grid.h:390-446 (interp<DIM> :52-139)."""
import numpy as np
import pytest

import edm_amd.hip as H
import edm_amd.workloads as W

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    H.require_gpu()
    yield


CASES = [
    dict(name="2d_periodic", lo=[0.0, 0.0], hi=[8.0, 6.0], sp=[0.05, 0.04], per=[1, 1], sg=[0.2, 0.15], bnd=None),
    dict(name="2d_odd_nodes", lo=[0.0, 0.0], hi=[3.7, 2.9], sp=[0.1, 0.1], per=[1, 1], sg=[0.25, 0.25], bnd=None),
    dict(name="3d_periodic", lo=[0.0] * 3, hi=[8.0] * 3, sp=[0.125] * 3, per=[1, 1, 1], sg=[0.25] * 3, bnd=None),
    dict(name="3d_uneven", lo=[0.0] * 3, hi=[4.1, 3.3, 2.9], sp=[0.1, 0.11, 0.13], per=[1, 1, 1], sg=[0.2] * 3, bnd=None),
    # a non-periodic sub-grid (skin) inside a periodic boundary: samples are remapped, the last blocks are never read
    dict(name="3d_subgrid_in_periodic_box", lo=[-0.3, 0.0, 0.0], hi=[3.3, 4.0, 4.0], sp=[0.1] * 3, per=[0, 1, 1], sg=[0.2] * 3,
         bnd=([0.0, 0.0, 0.0], [6.0, 4.0, 4.0], [1, 1, 1])),
]


def _pair(c):
    out = []
    for mode in (1, 0):
        g = H.Gauss.create(c["lo"], c["hi"], c["sp"], c["per"], 1, c["sg"])
        if c["bnd"]:
            g.set_boundary(*c["bnd"])
        g.set_lookup_replica(mode)
        out.append(g)
    return out


def _same_lookups(a, b, q, what):
    Ea, Da = a.get_value_deriv(q)
    Eb, Db = b.get_value_deriv(q)
    assert np.array_equal(Ea, Eb) and np.array_equal(Da, Db), what
    n = len(q)
    mask = (W.splitmix64(5, n) % np.uint64(3)).astype(np.int32)
    fa = W.uniform(6, n * 3).reshape(n, 3)
    fb = fa.copy()
    ea = a.update_forces(q, fa, mask, 1)
    eb = b.update_forces(q, fb, mask, 1)
    # (forces bit-identical; the energy is a sum over samples whose association differs between the two kernels)
    assert abs(ea - eb) <= 1e-12 * max(abs(ea), abs(eb)) and np.array_equal(fa, fb), what
    return Ea


@pytest.mark.parametrize("c", CASES, ids=lambda c: c["name"])
def test_replica_lookups_bit_identical(c, workdir):
    a, b = _pair(c)
    dim = a.dim
    lo, hi = np.array(c["lo"]), np.array(c["hi"])
    q = np.zeros((20000, 3))
    q[:, :dim] = lo + (W.uniform(11, 20000 * dim).reshape(-1, dim) * 1.4 - 0.2) * (hi - lo)
    # nodes themselves, the periodic seam and the last cell
    q[:64, :dim] = lo + np.floor(W.uniform(12, 64 * dim).reshape(-1, dim) * 20) * np.array(c["sp"])
    q[64:96, :dim] = hi - W.uniform(13, 32 * dim).reshape(-1, dim) * np.array(c["sp"]) * 1.5
    E0 = _same_lookups(a, b, q, "empty grid")
    assert not E0.any()
    assert a.lookup_replica_info()[0] and not b.lookup_replica_info()[0]
    assert a.lookup_replica_info()[1] == a.size * 128
    # short hill batches, several times: the in-place gather updates the replica -- no rebuild
    for k in range(4):
        hx = np.zeros((37 + 50 * k, 3))
        hx[:, :dim] = lo + (W.uniform(20 + k, len(hx) * dim).reshape(-1, dim) * 1.1 - 0.05) * (hi - lo)
        hh = (0.05 + W.uniform(30 + k, len(hx))) * (1 if k != 2 else -1)
        assert np.array_equal(a.add_values(hx, hh), b.add_values(hx, hh))
        E = _same_lookups(a, b, q, "after short batch %d" % k)
        assert E.any()
    assert a.lookup_replica_info() == (True, a.size * 128, 1), "short batches must not trigger a rebuild"
    va, da = a.download()
    vb, db = b.download()
    assert np.array_equal(va, vb) and np.array_equal(da, db)
    # a dense batch takes the grouped / fused application: replica stale, rebuilt by the next lookup
    n_dense = 6000
    hx = np.zeros((n_dense, 3))
    hx[:, :dim] = lo + W.uniform(40, n_dense * dim).reshape(-1, dim) * (hi - lo)
    assert np.array_equal(a.add_values(hx, 1e-3), b.add_values(hx, 1e-3))
    _same_lookups(a, b, q, "after a dense batch")
    builds = a.lookup_replica_info()[2]
    assert builds in (1, 2)   # (2 where the dense batch did not qualify for the in-place gather)
    # upload, Grid::add and clear all leave the replica stale
    a.upload(va * 0.5, da * 2.0)
    b.upload(vb * 0.5, db * 2.0)
    _same_lookups(a, b, q, "after upload")
    b.write(str(workdir / "other.grid"))
    other = H.Grid.read_file(dim, str(workdir / "other.grid"), 1)
    a.add(other, 0.7, 0.01)
    b.add(other, 0.7, 0.01)
    _same_lookups(a, b, q, "after Grid::add")
    a.clear()
    b.clear()
    assert not _same_lookups(a, b, q, "after clear").any()
    assert a.lookup_replica_info()[2] == builds + 3
    # switching it off frees it; on again rebuilds
    a.set_lookup_replica(0)
    assert a.lookup_replica_info()[:2] == (False, 0)
    a.set_lookup_replica(1)
    a.add_values(hx[:50], 0.3)
    b.add_values(hx[:50], 0.3)
    _same_lookups(a, b, q, "after re-enabling")


def test_walls_keep_the_node_records(workdir):
    """A grid with a non-periodic boundary dimension duplicates boundary values after hill batches
    (gaussian_grid.h:571-630); it stays on the node records even when the replica is requested."""
    g = H.Gauss.create([0.0, 0.0], [4.0, 4.0], [0.1, 0.1], [1, 0], 1, [0.2, 0.2])
    g.set_lookup_replica(1)
    g.add_values(np.array([[1.0, 0.05, 0.0]]), 1.0)
    E, _ = g.get_value_deriv(np.array([[1.0, 0.1, 0.0]]))
    assert E[0] > 0 and g.lookup_replica_info()[:2] == (False, 0)


@pytest.mark.parametrize("dim", [2, 3])
def test_controller_steps_with_replica(dim, workdir):
    """fix edm's hill-depositing step (edm_hip_bias_step: forces + add_hills, limiter active) with and without the
    replica: energies, forces, grids, histogram and limiter state bit for bit."""
    box = 8.0 if dim == 2 else 6.0
    text = ("tempering 0\nhill_prefactor 0.5\nhill_density 60\nbias_per_step 0.3\ndimension %d\nbox_low %s\nbox_high %s\n"
            "bias_spacing %s\nbias_sigma %s\n" % (dim, " ".join(["0"] * dim), " ".join(["%g" % box] * dim),
                                                  " ".join(["0.05" if dim == 2 else "0.1"] * dim), " ".join(["0.2"] * dim)))
    n = 30000
    runs = []
    for mode in (1, 0):
        cfg = str(workdir / ("m%d.edm" % mode))
        open(cfg, "w").write(text + "hills_filename %s/H%d\nhistogram_filename %s/HI%d\n" % (workdir, mode, workdir, mode))
        b = H.Bias(cfg)
        b.setup(1.0, 1.0)
        b.subdivide([0] * dim, [box] * dim, [0] * dim, [box] * dim, [1] * dim, [0] * dim)
        g = b.gauss
        g.set_lookup_replica(mode)
        out = []
        d_f = H.DeviceArray.zeros((n, 3))
        for step in range(6):
            x = W.atom_positions(n, 200 + step, box)
            d_x = H.DeviceArray.from_host(x)
            d_u = H.DeviceArray.from_host(W.uniform(300 + step, n))
            e = b.step_device(d_x, 3, d_f, 3, n, d_u)
            out.append((e, d_f.to_host(), [b.get(k) for k in ("cum_bias", "overflow_left", "overflow_right", "b_skip_hill_add", "hills_added")]))
        v, dv = g.download()
        runs.append((out, v, dv, b.hist.values, g.lookup_replica_info()))
        del b
    for (e1, f1, s1), (e2, f2, s2) in zip(runs[0][0], runs[1][0]):
        assert abs(e1 - e2) <= 1e-12 * abs(e2) and np.array_equal(f1, f2) and s1 == s2
    assert runs[0][0][-1][0] > 0
    for k in (1, 2, 3):
        assert np.array_equal(runs[0][k], runs[1][k])
    assert runs[0][4][0] and runs[0][4][2] == 1 and not runs[1][4][0]   # built once, then maintained by the gathers


@pytest.mark.parametrize("tag", ["c2d_2048sq", "c3d_512cube"])
def test_full_size_automatic_mode(tag):
    """BASELINE configs[3] / [4] grids: the automatic mode uses the replica (537 MB / 17.2 GB); forces of 262 144
    atoms equal those computed on the node records bit for bit, before and after culled hill batches."""
    c = W.C2D if tag.startswith("c2d") else W.C3D
    a = H.Gauss.create(c["lo"], c["hi"], c["spacing"], c["periodic"], 1, c["sigma"])
    b = H.Gauss.create(c["lo"], c["hi"], c["spacing"], c["periodic"], 1, c["sigma"])
    b.set_lookup_replica(0)
    n = 262144
    x = W.atom_positions(n, 21 if c["dim"] == 2 else 31)
    d_x = H.DeviceArray.from_host(x)
    for rnd in range(3):
        hills = x[rnd * 250:(rnd + 1) * 250].copy()
        assert np.array_equal(a.add_values(hills, 0.01), b.add_values(hills, 0.01))
        out = []
        for g in (a, b):
            d_f = H.DeviceArray.zeros((n, 3))
            e = H.C.c_double(0)
            H.check(H.lib().edm_hip_gauss_update_forces(g.h, n, d_x.ptr, 3, d_f.ptr, 3, None, -1, H.C.byref(e)))
            out.append((e.value, d_f.to_host()))
        assert abs(out[0][0] - out[1][0]) <= 1e-12 * abs(out[1][0]) and np.array_equal(out[0][1], out[1][1])
        assert np.abs(out[0][1]).max() > 0
    used, nbytes, builds = a.lookup_replica_info()
    assert used and nbytes == a.size * 128 and builds == 1
    assert not b.lookup_replica_info()[0]
