"""The grid factories and file readers of the public API on the HIP path (through the C ABI): read_grid /
Grid::read / Grid::write / Grid::add / set_interpolation (lib/grid.h:275-290, :343-446, :448-503, :712-835,
:911-928), read_gauss_grid and GaussGrid::read (lib/gaussian_grid.h:85-93, :140-142, :647), multi_write of a
plain grid with derivatives (grid.h:509-674) and DimmedGaussGrid::remap (:504-541) -- against the oracle on the
reference's own fixture grids (tests/golden/ref_fixtures/{1,2,3}.grid) and the reference's re-written files."""
import os

import numpy as np
import pytest

import edm_amd.hip as H
import edm_amd.workloads as W
from oracle import binding as B

import golden_util as GU

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    H.require_gpu()
    yield


def _queries(o, n, seed):
    lo, hi = o.min, o.max
    return lo + (W.uniform(seed, n * o.dim).reshape(n, o.dim) * 1.2 - 0.1) * (hi - lo)


@pytest.mark.parametrize("dim", [1, 2, 3])
def test_read_grid_write_and_lookup(dim, oracle_lib, workdir):
    src = os.path.join(GU.FIXTURES, "%d.grid" % dim)
    g = H.Grid.read_file(dim, src, 1)
    o = B.Grid.read(oracle_lib, dim, src, 1)
    # geometry: integers and host-computed doubles exact (grid.h:799-806)
    assert list(g.number) == list(o.number) and list(g.periodic) == list(o.periodic)
    assert list(g.dx) == list(o.dx) and list(g.min) == list(o.min) and list(g.max) == list(o.max)
    assert g.has_derivatives and o.has_deriv and g.size == o.size
    # contents: parsed text, sign of the derivative columns flipped (:828) -- exact
    assert np.array_equal(g.values, o.values) and np.array_equal(g.derivs, o.derivs)
    # DimmedGrid::write: byte-identical to the file the REFERENCE wrote after reading the same fixture
    out = str(workdir / ("%d.rewritten" % dim))
    g.write(out)
    assert open(out).read() == open(os.path.join(GU.GOLDEN, "file_%d_rewritten.grid" % dim)).read()
    # Grid::read of that file into the existing grid: same contents as the oracle doing the same
    o.write(str(workdir / "o.rewritten"))
    g.read(out)
    o2 = B.Grid.read(oracle_lib, dim, str(workdir / "o.rewritten"), 1)
    assert np.array_equal(g.values, o2.values) and np.array_equal(g.derivs, o2.derivs)
    # interpolated lookups (interp<DIM>, grid.h:52-139), in and outside in_grid
    q = _queries(o2, 4000, 60 + dim)
    E, D = g.get_value_deriv(q)
    ref = [o2.get_value_deriv(x) for x in q]
    rE, rD = np.array([r[0] for r in ref]), np.array([r[1] for r in ref])
    assert (rE == 0).sum() > 100 and (rE != 0).sum() > 1000
    assert np.allclose(E, rE, rtol=1e-10, atol=1e-13 * np.abs(rE).max())
    assert np.allclose(D, rD, rtol=1e-9, atol=1e-11 * np.abs(rD).max())
    assert np.array_equal(E == 0, rE == 0)
    # without interpolation: the nearest-lower node's value and stored derivatives, exactly
    g.set_interpolation(0)
    o2.set_interpolation(0)
    E, D = g.get_value_deriv(q)
    ref = [o2.get_value_deriv(x) for x in q]
    assert np.array_equal(E, np.array([r[0] for r in ref])) and np.array_equal(D, np.array([r[1] for r in ref]))
    if dim == 3:   # edm_test.cpp:117-125: the 3.grid known answer
        g.set_interpolation(1)
        v, _ = g.get_value_deriv([[0.75, 0.0, 1.00]])
        assert (v[0] - 1.260095) ** 2 < 1e-10


@pytest.mark.parametrize("dim", [1, 2, 3])
def test_grid_add_and_plain_multi_write(dim, oracle_lib, workdir):
    """Grid::add (grid.h:275-290) into a fresh make_grid(.., b_derivatives 1, b_interpolate 1) of a DIFFERENT spacing
    (so `other` is really interpolated), then DimmedGrid::multi_write of the result."""
    src = os.path.join(GU.FIXTURES, "%d.grid" % dim)
    other_g = H.Grid.read_file(dim, src, 1)
    other_o = B.Grid.read(oracle_lib, dim, src, 1)
    lo, hi = list(other_o.min), [float(v) for v in (other_o.max - other_o.dx * (1 - other_o.periodic))]
    sp = [float(v) * 0.37 for v in other_o.dx]
    per = [int(v) for v in other_o.periodic]
    g = H.Grid.create(lo, hi, sp, per, 1, 1)
    o = B.Grid.create(oracle_lib, lo, hi, sp, per, 1, 1)
    assert list(g.number) == list(o.number) and g.has_derivatives
    g.add(other_g, 1.5, 0.25)
    o.add_grid(other_o, 1.5, 0.25)
    vmax = np.abs(o.values).max()
    assert np.allclose(g.values, o.values, rtol=1e-11, atol=1e-14 * vmax)
    assert np.allclose(g.derivs, o.derivs, rtol=1e-10, atol=1e-12 * np.abs(o.derivs).max())
    # nodes outside in_grid of `other` receive only the offset
    assert np.array_equal(g.values == 0.25, o.values == 0.25)
    box_lo, box_hi = lo, hi
    g.multi_write(str(workdir / "g.mw"), box_lo, box_hi, per, 0)
    o.multi_write(str(workdir / "o.mw"), box_lo, box_hi, per, 0)
    a, w = open(str(workdir / "g.mw")).read().split("\n"), open(str(workdir / "o.mw")).read().split("\n")
    assert len(a) == len(w) and a[:7] == w[:7]
    diff = [(x, y) for x, y in zip(a, w) if x != y]
    # the body is 8-decimal text of interpolated doubles that agree to ~1e-15: at most a last-digit flip
    for x, y in diff:
        assert np.allclose([float(t) for t in x.split()], [float(t) for t in y.split()], rtol=0, atol=1.01e-8)
    assert len(diff) <= max(2, len(a) // 1000)
    # a histogram-flavoured grid (no derivatives) refuses to interpolate but accepts add_value; the other way round
    with pytest.raises(H.EdmHipError):
        g.add_values(H.DeviceArray.from_host(np.zeros((1, dim))), 1, dim)


def test_read_gauss_grid_vs_oracle(oracle_lib, workdir):
    """read_gauss_grid (gaussian_grid.h:647): rebuilt from 2.grid (PBC 0 1) with sigma given again; boundary = the
    grid's own extent; hills on top; GaussGrid::read into an existing grid keeps sigma and boundary."""
    src = os.path.join(GU.FIXTURES, "2.grid")
    sg = [0.12, 0.3]
    g = H.Gauss.read_file(2, src, sg)
    o = B.Gauss.read(oracle_lib, 2, src, sg)
    og = o.grid
    assert list(g.number) == list(og.number) and g.minisize == o.minisize
    assert list(g.boundary_min) == list(o.boundary_min) and list(g.boundary_max) == list(o.boundary_max)
    assert list(g.boundary_periodic) == list(o.boundary_periodic) and list(g.sigma) == list(o.sigma)
    v, dv = g.download()
    assert np.array_equal(v, og.values) and np.array_equal(dv, og.derivs)
    hx = np.zeros((40, 3))
    hx[:, :2] = og.min + W.uniform(71, 80).reshape(40, 2) * (og.max - og.min)
    hh = 0.1 + W.uniform(72, 40)
    added = g.add_values(hx, hh)
    ref = np.array([o.add_value(x[:2], float(h)) for x, h in zip(hx, hh)])
    assert np.allclose(added, ref, rtol=1e-10, atol=1e-15)
    v, dv = g.download()
    assert np.allclose(v, og.values, rtol=1e-10, atol=1e-13 * np.abs(og.values).max())
    assert np.allclose(dv, og.derivs, rtol=1e-10, atol=1e-12 * np.abs(og.derivs).max())
    g.write(str(workdir / "g2.grid"))
    g2 = H.Gauss.create([0.0, 0.0], [1.0, 1.0], [0.5, 0.5], [0, 0], 1, sg)
    g2.read(str(workdir / "g2.grid"))
    assert list(g2.number) == list(g.number) and list(g2.sigma) == list(g.sigma)
    assert list(g2.boundary_max) == [1.0, 1.0]   # the boundary of the grid it WAS is kept (gaussian_grid.h:140-142)
    v2, _ = g2.download()
    assert np.allclose(v2, v, rtol=0, atol=1.01e-8)
    # Grid::add with a GaussGrid as `other`: evaluated through its boundary-aware get_value_deriv
    p = H.Grid.create(list(og.min), [2.5, np.pi], [0.11, 0.13], [0, 1], 1, 1)
    po = B.Grid.create(oracle_lib, list(og.min), [2.5, np.pi], [0.11, 0.13], [0, 1], 1, 1)
    p.add(g, 1.0, 0.0)
    for i in range(po.size):   # the oracle has no Gauss-as-other entry: node by node
        x = po.min + po.dx * np.array(po.one2multi(i))
        e, d = o.get_value_deriv(x)
        po.values[i] += e
        po.derivs[i] += d
    assert np.allclose(p.values, po.values, rtol=1e-10, atol=1e-13 * np.abs(po.values).max())
    assert np.allclose(p.derivs, po.derivs, rtol=1e-9, atol=1e-11 * np.abs(po.derivs).max())


def test_remap_vs_oracle(oracle_lib):
    """DimmedGaussGrid::remap (gaussian_grid.h:504-541) as the device code applies it, bit for bit: a periodic grid,
    and a non-periodic sub-grid inside a periodic boundary (edm_test.cpp:252-333)."""
    cases = [
        dict(lo=[0, 0], hi=[10, 5], sp=[1, 1], per=[1, 0], sg=[0.1, 0.1], bnd=([0, 0], [10, 10], [1, 1])),
        dict(lo=[-2], hi=[7], sp=[0.1], per=[0], sg=[0.1], bnd=([0], [10], [1])),
        dict(lo=[-2], hi=[7], sp=[0.1], per=[0], sg=[0.1], bnd=([0], [10], [0])),
        dict(lo=[0.0] * 3, hi=[4, 5, 6], sp=[0.5] * 3, per=[1, 0, 1], sg=[0.3] * 3, bnd=([0, -5, 0], [4, 10, 6], [1, 1, 1])),
    ]
    for k, c in enumerate(cases):
        g = H.Gauss.create(c["lo"], c["hi"], c["sp"], c["per"], 1, c["sg"])
        o = B.Gauss.create(oracle_lib, c["lo"], c["hi"], c["sp"], c["per"], 1, c["sg"])
        g.set_boundary(*c["bnd"])
        o.set_boundary(*c["bnd"])
        dim = len(c["lo"])
        lo, hi = np.array(c["lo"], float), np.array(c["hi"], float)
        q = lo + (W.uniform(80 + k, 3000 * dim).reshape(-1, dim) * 5 - 2) * (hi - lo)
        q[:6] = np.array([0, 1, -1, 6, 11, 9])[:, None]   # the reference test's own points
        got = g.remap(q)
        want = np.array([o.remap(x) for x in q])
        assert np.array_equal(got, want), "case %d" % k


@pytest.mark.parametrize("lammps", [0, 1], ids=["plumed_layout", "lammps_table"])
def test_multi_write_of_a_skin_offset_subgrid_vs_oracle(lammps, oracle_lib, workdir):
    """DimmedGrid::multi_write (grid.h:509-674) where it is NOT a node dump: the pair fix's grid reaches a skin beyond
    the boundary on both sides (fix_edm_pair.cpp:96-104 -> edm_bias.cpp:142-154: grid [-0.3, 3.6], boundary [1, 3.3]), so
    the rows box_min + k dx fall between nodes and are re-sampled by interpolation; the LAMMPS table prepends its
    filler rows.  Hills near both walls (McGovern-De Pablo terms active).  Text equal to the oracle's up to last-digit
    flips of the 8-decimal body (device exp vs libm), header and row structure exact."""
    lo, hi, sp, sg = [-0.3], [3.6], [0.24375], [0.3]
    g = H.Gauss.create(lo, hi, sp, [0], 1, sg)
    o = B.Gauss.create(oracle_lib, lo, hi, sp, [0], 1, sg)
    g.set_boundary([1.0], [3.3], [0])
    o.set_boundary([1.0], [3.3], [0])
    hx = np.zeros((30, 3))
    hx[:, 0] = 1.0 + W.uniform(91, 30) * 2.3
    hx[:3, 0] = [1.0, 3.3, 1.02]
    hh = 0.2 + W.uniform(92, 30)
    g.add_values(hx, hh)
    for x, h in zip(hx, hh):
        o.add_value(x[:1], float(h))
    g.multi_write(str(workdir / "g.out"), lammps)
    o.multi_write(str(workdir / "o.out"), lammps)
    a, w = open(str(workdir / "g.out")).read().split("\n"), open(str(workdir / "o.out")).read().split("\n")
    assert len(a) == len(w) and len(a) > 12
    nhead = 7 if not lammps else 4
    assert a[:nhead] == w[:nhead], "header"
    flips = 0
    for x, y in zip(a[nhead:], w[nhead:]):
        if x == y:
            continue
        tx, ty = x.split(), y.split()
        assert len(tx) == len(ty) and tx[0] == ty[0], (x, y)
        assert np.allclose([float(t) for t in tx], [float(t) for t in ty], rtol=0, atol=1.01e-8), (x, y)
        flips += 1
    assert flips <= 2, "%d rows differ in a last digit" % flips
