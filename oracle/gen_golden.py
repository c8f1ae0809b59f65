#!/usr/bin/env python3
"""Generates tests/golden/*.json|npz|txt by running the REAL reference
(oracle/_ref/libedm_ref.so, built from /root/reference/lib by oracle/Makefile)
on seeded inputs.  Fixtures hold inputs and expected outputs only -- numbers and
the text files the reference writes -- never reference source.

Run in the build container (where /root/reference exists):
    python oracle/gen_golden.py
TEST INFRASTRUCTURE ONLY.
"""
import hashlib
import json
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "electronic-dance-music_amd"))

from oracle import binding as B  # noqa: E402
import workloads as W  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tests"))
import pairfix_cases as PF  # noqa: E402  (seeded inputs shared with the tests)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def u(seed, n):
    return W.uniform(seed, n)


# --------------------------------------------------------------------------
# scenario 1: gaussian-grid hills + interpolated lookups
# --------------------------------------------------------------------------
GAUSS_SCENARIOS = [
    # name, geometry, n_hills, n_queries
    dict(name="c1d_full", lo=[0.0], hi=[2.8], sp=[0.00025], per=[0], sg=[0.025], bnd=None, nh=64, nq=2048),
    dict(name="c1d_skin", lo=[-0.3], hi=[3.6], sp=[0.002], per=[0], sg=[0.04], bnd=([0.0], [3.3], [0]), nh=48, nq=512),
    dict(name="p1d", lo=[-np.pi], hi=[np.pi], sp=[np.pi / 200], per=[1], sg=[0.05], bnd=None, nh=48, nq=512),
    dict(name="sub1d_in_periodic_box", lo=[-2.0], hi=[7.0], sp=[0.1], per=[0], sg=[0.3], bnd=([0.0], [10.0], [1]), nh=32, nq=256),
    dict(name="mcgdp_inside_1d", lo=[-100.0], hi=[100.0], sp=[1.0], per=[1], sg=[10.0], bnd=([-50.0], [50.0], [0]), nh=20, nq=256),
    dict(name="c2d_small", lo=[0.0, 0.0], hi=[8.0, 8.0], sp=[1 / 8.0, 1 / 8.0], per=[1, 1], sg=[0.25, 0.25], bnd=None, nh=40, nq=512),
    dict(name="m2d_mixed", lo=[0.0, 0.0], hi=[8.0, 6.0], sp=[0.25, 0.2], per=[1, 0], sg=[0.5, 0.4], bnd=None, nh=32, nq=512),
    dict(name="n2d_mcgdp", lo=[0.0, 0.0], hi=[8.0, 6.0], sp=[0.25, 0.2], per=[0, 0], sg=[0.5, 0.4], bnd=None, nh=32, nq=512),
    dict(name="c3d_small", lo=[0.0] * 3, hi=[3.0] * 3, sp=[0.125] * 3, per=[1, 1, 1], sg=[0.25] * 3, bnd=None, nh=16, nq=512),
    dict(name="n3d_mcgdp_inside", lo=[-10.0] * 3, hi=[10.0] * 3, sp=[0.9, 1.1, 1.4], per=[1, 1, 1], sg=[3.0] * 3,
         bnd=([-5.0] * 3, [5.0] * 3, [0, 0, 0]), nh=12, nq=256),
]


def scenario_inputs(sc, idx):
    dim = len(sc["lo"])
    blo = np.array(sc["bnd"][0] if sc["bnd"] else sc["lo"], dtype=float)
    bhi = np.array(sc["bnd"][1] if sc["bnd"] else sc["hi"], dtype=float)
    seed = 1000 + 10 * idx
    hx = blo + u(seed, sc["nh"] * dim).reshape(sc["nh"], dim) * (bhi - blo)
    hh = 0.2 + u(seed + 1, sc["nh"])
    hh[::7] *= -0.5  # negative heights occur as "undo" hills
    # a few special positions: on the boundary, just outside, one period away
    hx[0] = blo
    hx[1] = bhi
    hx[2] = blo - 1e-9
    hx[3] = bhi + 0.37 * (bhi - blo)
    glo = np.minimum(np.array(sc["lo"], dtype=float), blo)
    ghi = np.maximum(np.array(sc["hi"], dtype=float), bhi)
    q = glo + (u(seed + 2, sc["nq"] * dim).reshape(sc["nq"], dim) * 1.4 - 0.2) * (ghi - glo)
    q[0] = blo
    q[1] = bhi
    return hx, hh, q


def gen_gauss(lib):
    meta = []
    for idx, sc in enumerate(GAUSS_SCENARIOS):
        g = B.Gauss.create(lib, sc["lo"], sc["hi"], sc["sp"], sc["per"], 1, sc["sg"])
        if sc["bnd"]:
            g.set_boundary(*sc["bnd"])
        hx, hh, q = scenario_inputs(sc, idx)
        dim = g.dim
        added = np.array([g.add_value(x, float(h)) for x, h in zip(hx, hh)])
        E = np.zeros(len(q))
        der = np.zeros((len(q), dim))
        flat = np.full(len(q), -1, dtype=np.int64)
        gg = g.grid
        for i, x in enumerate(q):
            E[i], der[i] = g.get_value_deriv(x)
            xr = x.copy()
            if not g.in_bounds(xr):
                xr = g.remap(xr)
            if g.in_bounds(xr) and gg.in_grid(xr):
                flat[i] = gg.multi2one(gg.get_index(xr))
        tabs = {}
        for d in range(dim):
            if not g.boundary_periodic[d]:
                t0, t1 = g.bc_table(d, 0), g.bc_table(d, 1)
                tabs[str(d)] = dict(
                    sha256=hashlib.sha256(t0.tobytes() + t1.tobytes()).hexdigest(),
                    sample_index=list(range(0, 65536, 4099)),
                    denom=[float(v) for v in t0[::4099]],
                    dderiv=[float(v) for v in t1[::4099]],
                )
        np.savez_compressed(
            os.path.join(GOLDEN, "gauss_%s.npz" % sc["name"]),
            hill_x=hx, hill_h=hh, queries=q, bias_added=added, E=E, der=der, flat_index=flat,
            grid_values=gg.values.copy(), grid_derivs=gg.derivs.copy(),
        )
        meta.append(dict(
            name=sc["name"], lo=list(map(float, sc["lo"])), hi=list(map(float, sc["hi"])),
            sp=list(map(float, sc["sp"])), per=sc["per"], sg=list(map(float, sc["sg"])),
            bnd=[list(map(float, sc["bnd"][0])), list(map(float, sc["bnd"][1])), sc["bnd"][2]] if sc["bnd"] else None,
            grid_number=[int(v) for v in gg.number], dx=[float(v) for v in gg.dx],
            grid_max=[float(v) for v in gg.max], minisize=g.minisize, sigma_eff=[float(v) for v in g.sigma],
            tables=tabs,
        ))
    with open(os.path.join(GOLDEN, "gauss_scenarios.json"), "w") as fh:
        json.dump(meta, fh, indent=1)


# --------------------------------------------------------------------------
# scenario 2: the bias controller (heights, limiting, undo, overflow buffer)
# --------------------------------------------------------------------------
CTRL = {
    "limit_unit_hills": dict(
        cfg="tempering 0\nhill_prefactor 1.0\nbias_per_step 1.5\ndimension 1\nbox_low 0\nbox_high 2.8\n"
            "bias_spacing 0.001\nbias_sigma 0.05",
        per=[0], skin=[0.3], mode="explicit", steps=5, n=6),
    "density_c1d": dict(
        cfg="tempering 0\nhill_prefactor 0.5\nhill_density 40\ndimension 1\nbox_low 0\nbox_high 2.8\n"
            "bias_spacing 0.00025\nbias_sigma 0.025",
        per=[0], skin=[0.3], mode="array", steps=4, n=4096),
    "all_samples_c1d": dict(
        cfg="tempering 0\nhill_prefactor 0.5\nbias_per_step 100.0\ndimension 1\nbox_low 0\nbox_high 2.8\n"
            "bias_spacing 0.001\nbias_sigma 0.05",
        per=[0], skin=[0.3], mode="array", steps=3, n=512),
    "local_tempering": dict(
        cfg="tempering 1\nbias_factor 10\nglobal_tempering -1\nhill_prefactor 0.02\nbias_per_step 5.0\n"
            "dimension 1\nbox_low 0\nbox_high 2.8\nbias_spacing 0.001\nbias_sigma 0.05",
        per=[0], skin=[0.3], mode="array", steps=3, n=256),
    "global_tempering": dict(
        cfg="tempering 1\nbias_factor 5\nglobal_tempering 0.2\nhill_prefactor 0.4\ndimension 1\nbox_low 0\n"
            "box_high 2.8\nbias_spacing 0.001\nbias_sigma 0.05",
        per=[0], skin=[0.3], mode="array", steps=6, n=64),
    "targeting_global_tempering": dict(
        cfg="tempering 1\nbias_factor 5.0\nglobal_tempering 0.05\nhill_prefactor 0.2\nbias_per_step 0.15\nhill_density 30\n"
            "target_filename @FIXTURES@/1.grid\ndimension 1\nbox_low 0\nbox_high 2.5\nbias_spacing 0.0025\nbias_sigma 0.05",
        per=[0], skin=[0.3], mode="array", steps=6, n=2000),
    "density_2d_limit": dict(
        cfg="tempering 0\nhill_prefactor 0.3\nhill_density 10\nbias_per_step 0.12\ndimension 2\nbox_low 0 0\n"
            "box_high 8 8\nbias_spacing 0.25 0.25\nbias_sigma 0.5 0.4",
        per=[1, 1], skin=[0.0, 0.0], mode="array", steps=5, n=400),
    "density_3d_limit": dict(
        cfg="tempering 0\nhill_prefactor 0.02\nhill_density 8\nbias_per_step 0.008\ndimension 3\nbox_low 0 0 0\n"
            "box_high 3 3 3\nbias_spacing 0.125 0.125 0.125\nbias_sigma 0.25 0.25 0.25",
        per=[1, 1, 1], skin=[0.0] * 3, mode="array", steps=4, n=300),
}


def ctrl_inputs(name, spec, step, dim, lo, hi):
    seed = 5000 + 100 * sorted(CTRL).index(name) + step
    n = spec["n"]
    pos = np.zeros((n, 3))
    pos[:, :dim] = lo + u(seed, n * dim).reshape(n, dim) * (hi - lo) * 1.04 - 0.02 * (hi - lo)
    ru = u(seed + 50, n)
    mask = (W.splitmix64(seed + 77, n) % np.uint64(4)).astype(np.int32)
    return pos, ru, mask


def run_controller(lib, name, spec, tmp):
    cfg = os.path.join(tmp, name + ".edm")
    hills = os.path.join(tmp, "HILLS_" + name)
    with open(cfg, "w") as fh:
        fh.write(spec["cfg"].replace("@FIXTURES@", os.path.join(GOLDEN, "ref_fixtures"))
                 + "\nhills_filename %s\nhistogram_filename %s.hist\n" % (hills, hills))
    b = B.Bias(lib, cfg)
    dim = int(b.get("dim"))
    b.setup(1.0, 1.0)
    lo, hi = b.array("min"), b.array("max")
    b.subdivide(lo, hi, lo, hi, spec["per"], spec["skin"])
    rec = dict(E=[], cum=[], overflow=[], forces_sha=[], forces_head=[])
    for step in range(spec["steps"]):
        pos, ru, mask = ctrl_inputs(name, spec, step, dim, lo, hi)
        forces = np.zeros_like(pos)
        apply_mask = 1 if step % 2 else -1
        b.set_mask(mask)
        rec["E"].append(b.update_forces(pos, forces, apply_mask))
        rec["forces_sha"].append(hashlib.sha256(forces.tobytes()).hexdigest())
        rec["forces_head"].append(forces[:8, :dim].tolist())
        if spec["mode"] == "explicit":
            b.pre_add_hill(1)
            for k in range(spec["n"]):
                b.add_hill(pos[k], float(ru[k]))
            b.post_add_hill()
        else:
            b.add_hills(pos, ru, apply_mask)
        rec["cum"].append(b.get("cum_bias"))
        rec["overflow"].append([int(b.get("overflow_left")), int(b.get("overflow_right")), int(b.get("b_skip_hill_add"))])
    gg = b.gauss.grid
    np.savez_compressed(os.path.join(GOLDEN, "ctrl_%s.npz" % name), grid_values=gg.values.copy(),
                        grid_derivs=gg.derivs.copy(), hist=b.hist.values.copy())
    bias_file = os.path.join(tmp, "BIAS_" + name)
    b.write_bias(bias_file)
    b.write_histogram()
    del b
    shutil.copy(hills + "_0", os.path.join(GOLDEN, "ctrl_%s.hills.txt" % name))
    if gg.size <= 4096:
        shutil.copy(bias_file, os.path.join(GOLDEN, "ctrl_%s.bias.grid" % name))
    shutil.copy(hills + ".hist", os.path.join(GOLDEN, "ctrl_%s.hist.grid" % name))
    rec.update(name=name, cfg=spec["cfg"], per=spec["per"], skin=spec["skin"], mode=spec["mode"],
               steps=spec["steps"], n=spec["n"], total_volume=None)
    return rec


def gen_controller(lib):
    tmp = tempfile.mkdtemp(prefix="edm_golden_")
    out = [run_controller(lib, name, CTRL[name], tmp) for name in sorted(CTRL)]
    with open(os.path.join(GOLDEN, "controller.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    shutil.rmtree(tmp)


# --------------------------------------------------------------------------
# scenario 3: known answers and file formats
# --------------------------------------------------------------------------
def gen_kats_and_files(lib):
    tmp = tempfile.mkdtemp(prefix="edm_golden_")
    fx = os.path.join(GOLDEN, "ref_fixtures")
    kat = {}
    # (i) notebook known answer (python-example/EDM.ipynb:103)
    cfg = os.path.join(tmp, "nb.edm")
    open(cfg, "w").write(open(os.path.join(fx, "notebook_input.edm")).read() + "\nhills_filename %s/H1\n" % tmp)
    b = B.Bias(lib, cfg)
    b.setup(1, 1)
    b.subdivide([0], [10], [0], [10], [0], [0])
    b.pre_add_hill(1)
    b.add_hill([0.25], 1.0)
    b.post_add_hill()
    v, d = b.gauss.get_value_deriv([0.24])
    kat["notebook"] = dict(value=v, deriv=float(d[0]), cum_bias=b.get("cum_bias"),
                           published=[1.1002417338159258, -0.6144025830861709])
    b.gauss.multi_write(os.path.join(GOLDEN, "file_notebook_multiwrite.grid"), 0)
    b.gauss.multi_write(os.path.join(GOLDEN, "file_notebook_lammps.ltab"), 1)
    del b
    # (ii) edm_sanity (tests/edm_test.cpp:873-887)
    cfg = os.path.join(tmp, "sanity.edm")
    open(cfg, "w").write(open(os.path.join(fx, "sanity.edm")).read() + "\nhills_filename %s/H2\n" % tmp)
    b = B.Bias(lib, cfg)
    b.setup(1, 1)
    b.subdivide([0], [10], [0], [10], [1], [0])
    b.add_hills(np.array([[5.0]]), [1.0])
    kat["sanity"] = dict(value_at_5=b.gauss.get_value([5.0]), cum_bias=b.get("cum_bias"), grid_size=b.gauss.grid.size,
                         d_left=float(b.gauss.get_value_deriv([4.99])[1][0]),
                         d_right=float(b.gauss.get_value_deriv([5.01])[1][0]))
    b.write_bias(os.path.join(GOLDEN, "file_sanity_bias.grid"))
    del b
    # (iii) 3.grid known value (tests/edm_test.cpp:117-125) and re-written fixtures
    for dim in (1, 2, 3):
        g = B.Grid.read(lib, dim, os.path.join(fx, "%d.grid" % dim), 1)
        g.write(os.path.join(GOLDEN, "file_%d_rewritten.grid" % dim))
        if dim == 3:
            g.set_interpolation(0)
            kat["grid3_nearest"] = dict(x=[0.75, 0.0, 1.0], value=g.get_value([0.75, 0.0, 1.0]), expected=1.260095)
            g.set_interpolation(1)
            v1, d1 = g.get_value_deriv([0.76, 0.0, 1.0])
            kat["grid3_interp"] = dict(x=[0.76, 0.0, 1.0], value=v1, deriv=d1.tolist())
    # (iv) geometry facts (tests/edm_test.cpp:25-107)
    g = B.Grid.create(lib, [-2, -5, -3], [125, 63, 78], [1.27, 1.36, 0.643], [0, 1, 1], 0, 0)
    kat["grid_3d_sanity_numbers"] = [int(v) for v in g.number]
    g = B.Grid.create(lib, [0], [10], [1], [0], 0, 0)
    kat["grid_1d_sanity_numbers"] = [int(v) for v in g.number]
    with open(os.path.join(GOLDEN, "kats.json"), "w") as fh:
        json.dump(kat, fh, indent=1)
    shutil.rmtree(tmp)


# --------------------------------------------------------------------------
# scenario 4: the pair fix's post_force in the REFERENCE'S order (lammps/fix_edm_pair.cpp:173-247):
# per pair update_force, then one or two add_hill -- pair k's force sees the hills of pairs 0..k-1
# --------------------------------------------------------------------------
def run_pairfix(lib, name, spec, tmp):
    cfg = os.path.join(tmp, name + ".edm")
    hills = os.path.join(tmp, "HILLS_pf_" + name)
    with open(cfg, "w") as fh:
        fh.write(spec["cfg"] + "\nhills_filename %s\nhistogram_filename %s.hist\n" % (hills, hills))
    b = B.Bias(lib, cfg)
    b.setup(1.0, 1.0)
    # FixEDMPair::init (fix_edm_pair.cpp:88-106): every rank's bounds are [0, cut + skin], not periodic
    b.subdivide([spec["lo"]], [spec["hi"]], [spec["lo"]], [spec["hi"]], [0], [spec["skin"]])
    last_calls = spec["nmax"]
    E, F, NC, CUM, OVF = [], [], [], [], []
    for step, hill_step in enumerate(spec["steps"]):
        r, second, ru = PF.pairfix_inputs(name, step)
        e, f, nc = b.pair_loop(r, second, ru, hill_step, last_calls)
        if hill_step:
            last_calls = nc
        E.append(e)
        F.append(f)
        NC.append(nc)
        CUM.append(b.get("cum_bias"))
        OVF.append([int(b.get("overflow_left")), int(b.get("overflow_right")), int(b.get("b_skip_hill_add"))])
    gg = b.gauss.grid
    np.savez_compressed(os.path.join(GOLDEN, "pairfix_%s.npz" % name), energy=np.array(E), force=np.array(F),
                        ncalls=np.array(NC, dtype=np.int64), cum_bias=np.array(CUM),
                        overflow=np.array(OVF, dtype=np.int64), grid_values=gg.values.copy(),
                        grid_derivs=gg.derivs.copy(), hist=b.hist.values.copy())
    del b
    shutil.copy(hills + "_0", os.path.join(GOLDEN, "pairfix_%s.hills.txt" % name))


def gen_pairfix(lib):
    tmp = tempfile.mkdtemp(prefix="edm_golden_")
    for name in sorted(PF.PAIRFIX):
        run_pairfix(lib, name, PF.PAIRFIX[name], tmp)
    shutil.rmtree(tmp)


def main():
    lib = B.load("ref")
    os.makedirs(GOLDEN, exist_ok=True)
    gen_gauss(lib)
    gen_controller(lib)
    gen_kats_and_files(lib)
    gen_pairfix(lib)
    print("golden fixtures written to", GOLDEN)


if __name__ == "__main__":
    main()
