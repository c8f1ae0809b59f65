"""Randomised (seeded) parity sweep of the reference-order fix edm_pair step (edm_hip_bias_pair_step_ordered: record and
force pass on the object's second stream, beside the hill batch's launch) against the CPU oracle driven in the
reference fix's own loop (oracle/edm_oracle.c:ora_bias_pair_loop <- lammps/fix_edm_pair.cpp:173-247).  Drawn: grid
size and hill width, walls on the grid's edge or strictly inside it (boundary duplication on real nodes, pairs beyond
the walls), tempering (none / global / local), stochastic and all-samples deposition, limiter pressure from none to
binding on every step, one or two add_hill calls per pair, pair counts from a handful to tens of thousands (odd ones
among them), hill steps and steps without hills in turn."""
import os

import numpy as np
import pytest

import edm_amd.hip as H
import edm_amd.workloads as W
from oracle import binding as B

pytestmark = pytest.mark.gpu


def scenarios():
    # (EDM_FUZZ_SEED / EDM_FUZZ_COUNT in the environment: extended sweeps beyond the committed configurations)
    rng = np.random.default_rng(int(os.environ.get("EDM_FUZZ_SEED", "4242")))
    out = []
    for k in range(int(os.environ.get("EDM_FUZZ_COUNT", "20"))):
        hi = float(rng.uniform(2.0, 4.0))
        nodes = int(rng.integers(200, 6000))
        sp = hi / nodes
        sg = float(rng.uniform(2.0, 12.0)) * sp
        inside = bool(rng.random() < 0.4)          # walls strictly inside the grid
        b_lo = float(rng.uniform(0.1, 0.6)) if inside else 0.0
        b_hi = hi - float(rng.uniform(0.05, 0.4)) if inside else hi
        mode = str(rng.choice(["plain", "plain", "plain", "global", "local"]))
        stochastic = bool(rng.random() < 0.75) and mode != "local"
        n = int(rng.choice([7, 301, 4097, 30000])) if stochastic else int(rng.choice([5, 120, 700]))
        if mode == "local":
            n = int(rng.choice([5, 90, 400]))
        density = float(rng.uniform(5, 200)) if stochastic else None
        prefactor = float(rng.uniform(0.02, 0.6))
        limit = float(rng.uniform(0.2, 2.5)) * prefactor if rng.random() < 0.7 else None
        cfg = "tempering %d\n" % (0 if mode == "plain" else 1)
        if mode == "global":
            cfg += "bias_factor %g\nglobal_tempering %g\n" % (rng.uniform(2, 12), rng.uniform(0.02, 0.5))
        if mode == "local":
            cfg += "bias_factor %g\nglobal_tempering -1\n" % rng.uniform(2, 12)
        cfg += "hill_prefactor %.6g\n" % prefactor
        if density:
            cfg += "hill_density %.6g\n" % density
        if limit:
            cfg += "bias_per_step %.6g\n" % limit
        cfg += "dimension 1\nbox_low %.8g\nbox_high %.8g\nbias_spacing %.8g\nbias_sigma %.8g\n" % (b_lo, b_hi, sp, sg)
        steps = [1, 1, 0, 1, 1] if k % 3 else [1, 0, 1, 1]
        out.append(dict(name="%02d_%s_%s_n%d%s%s" % (k, mode, "dens" if density else "all", n, "_lim" if limit else "",
                                                       "_walls" if inside else ""),
                        cfg=cfg, lo=0.0, hi=hi, skin=float(rng.uniform(0.0, 0.3)), n=n, steps=steps,
                        seed=int(rng.integers(1, 1 << 30)), p_second=float(rng.choice([0.0, 0.6, 1.0]))))
    return out


@pytest.mark.parametrize("sc", scenarios(), ids=lambda s: s["name"])
def test_random_reference_order_steps_vs_oracle(sc, tmp_path):
    H.require_gpu()
    lib = B.load("oracle")
    handles = []
    for tag, cls in (("gpu", None), ("ora", lib)):
        cfg = str(tmp_path / (tag + ".edm"))
        open(cfg, "w").write(sc["cfg"] + "hills_filename %s/HILLS_%s\nhistogram_filename %s/HIST_%s\n" % (tmp_path, tag, tmp_path, tag))
        b = H.Bias(cfg) if cls is None else B.Bias(cls, cfg)
        b.setup(1.0, 1.0)
        b.subdivide([sc["lo"]], [sc["hi"]], [sc["lo"]], [sc["hi"]], [0], [sc["skin"]])
        handles.append(b)
    g, o = handles
    n = sc["n"]
    last = 2 * n
    for step, hill_step in enumerate(sc["steps"]):
        u = W.uniform(sc["seed"] + 10 * step, n)
        r = sc["lo"] - 0.1 + u * (sc["hi"] - sc["lo"] + 0.2)           # (some pairs beyond the grid / the walls)
        second = (W.uniform(sc["seed"] + 10 * step + 1, n) < sc["p_second"]).astype(np.int32)
        ru = W.uniform(sc["seed"] + 10 * step + 2, 2 * n)
        eo, fo, ncalls = o.pair_loop(r, second, ru, hill_step, last)
        d_r = H.DeviceArray.from_host(r)
        d_f = H.DeviceArray.from_host(np.zeros(n))
        if hill_step:
            reps = 1 + second
            xs, us = np.repeat(r, reps), ru[:int(reps.sum())].copy()
            first = np.zeros(n, dtype=np.int32)
            first[1:] = np.cumsum(reps[:-1])
            assert len(xs) == ncalls
            eg = g.pair_step_ordered_device(d_r, d_f, H.DeviceArray.from_host(first), n, H.DeviceArray.from_host(xs),
                                            H.DeviceArray.from_host(us), len(xs), est=last)
            last = len(xs)
        else:
            eg = g.pair_forces_device(d_r, d_f, n)
        fg = d_f.to_host()
        scale = max(np.abs(fo).max(), 1e-300)
        bad = np.abs(fg - fo) > 1e-8 * np.abs(fo) + 1e-9 * scale
        assert not bad.any(), "step %d: %d/%d forces differ, worst %g of %g" % (step, bad.sum(), n, np.abs(fg - fo).max(), scale)
        assert abs(eg - eo) <= 1e-9 * max(abs(eo), 1e-300) + 1e-12, (step, eg, eo)
        keys = ("overflow_left", "overflow_right", "b_skip_hill_add", "hills_added", "steps")
        assert [g.get(k) for k in keys] == [o.get(k) for k in keys], (step, [g.get(k) for k in keys], [o.get(k) for k in keys])
        assert abs(g.get("cum_bias") - o.get("cum_bias")) <= 1e-9 * max(abs(o.get("cum_bias")), 1e-300)
    v, dv = g.gauss.download()
    ov = o.gauss.grid.values
    assert np.allclose(v, ov, rtol=1e-9, atol=1e-12 * max(np.abs(ov).max(), 1e-300))
    assert np.array_equal(g.hist.values, o.hist.values)
    assert g.get("ord_gate_giveups") == 0
