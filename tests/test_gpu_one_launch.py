"""A hill-depositing fix edm_pair step as ONE launch (k_pair_step, opt-in: selection with per-hill integrals | pair
forces | bookkeeper | gather tiles, the grid written to its second buffer) against the same step as the library queues
it by default (k_pair_forces_select, then k_integrals_gather): every result bit for bit -- energies, forces, grid, gradient, histogram, controller state, HILLS log.  The reference loop
both replace is fix_edm_pair.cpp:174-246 over edm_bias.cpp:401-583."""
import numpy as np
import pytest

import edm_amd.hip as H
import edm_amd.workloads as W
from oracle import binding as B

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    H.require_gpu()
    yield

BASE = ("tempering 0\nhill_prefactor 0.5\ndimension 1\nbox_low 0\nbox_high 2.8\n")
TEMPERED = ("tempering 1\nbias_factor 5\nglobal_tempering 0.05\nhill_prefactor 0.5\ndimension 1\nbox_low 0\nbox_high 2.8\n")

# (a step runs as one launch when its selection workgroups -- 4096 samples each -- expect at most four accepted samples:
#  hill_density * 4096 / n <= 4)
CASES = {
    # limiter never binds: the tiles decide by themselves that nobody has to wait for it
    "below_limit": dict(cfg="hill_density 60\nbias_per_step 50\nbias_spacing 0.001\nbias_sigma 0.05\n", n=100_000, ns=None, steps=5),
    # limiter binds every step: the tiles wait for the bookkeeper's word, deferred hills, overflow flushes
    "limiter_binds": dict(cfg="hill_density 40\nbias_per_step 0.12\nbias_spacing 0.001\nbias_sigma 0.05\n", n=60_000, ns=None, steps=6),
    # an odd number of pairs, hill samples from another (shorter) array, a partial last selection chunk
    "other_samples": dict(cfg="hill_density 50\nbias_per_step 0.4\nbias_spacing 0.001\nbias_sigma 0.05\n", n=90_001, ns=70_000, steps=5),
    # BASELINE configs[1] geometry at its full size: 1 M pairs, 11 201 nodes, hill_density 250
    "w1_full": dict(cfg="hill_density 250\nbias_per_step 0.5\nbias_spacing 0.00025\nbias_sigma 0.025\n", n=1 << 20, ns=None, steps=4),
    # more than one 256-hill chunk per tile (the launch bound allows 2048 hills)
    "many_hills": dict(cfg="hill_density 400\nbias_per_step 2.0\nbias_spacing 0.001\nbias_sigma 0.05\n", n=500_000, ns=None, steps=3),
    # globally tempered heights (the prefactor shrinks with the accumulated bias, edm_bias.cpp:419-424: a new constant height
    # every step) on the production grid, walls included
    "global_tempering": dict(base=TEMPERED, cfg="hill_density 120\nbias_per_step 0.9\nbias_spacing 0.00025\nbias_sigma 0.025\n",
                             n=300_000, ns=None, steps=6),
    # one selection workgroup accepts more samples than it has slots for: the whole step falls back (synchronous redo)
    "slots_overflow": dict(cfg="hill_density 60\nbias_per_step 0.6\nbias_spacing 0.001\nbias_sigma 0.05\n", n=100_000, ns=None, steps=4,
                           clustered=True),
}


def run(tag, case, workdir, mode):
    cfg = str(workdir / (tag + ".edm"))
    open(cfg, "w").write(case.get("base", BASE) + case["cfg"] + "hills_filename %s/HILLS_%s\nhistogram_filename %s/HIST_%s\n" % (workdir, tag, workdir, tag))
    b = H.Bias(cfg)
    b.set("debug_pair_step_mode", mode)   # 0: forces+selection | integrals+gather (the default); 1: one launch
    b.setup(1.0, 1.0)
    b.subdivide([0], [2.8], [0], [2.8], [0], [0.3])
    n = case["n"]
    ns = case["ns"] or n
    energies, forces, states = [], [], []
    for step in range(case["steps"]):
        r = W.pair_distances(n, 4100 + step)
        rs = r if ns == n else W.pair_distances(ns, 4200 + step)
        u = W.uniform(4300 + step, ns)
        if case.get("clustered") and step == 1:
            u[100:140] = 0.0   # forty accepted samples inside one selection chunk
        d_r = H.DeviceArray.from_host(r)
        d_s = d_r if ns == n else H.DeviceArray.from_host(rs)
        d_u = H.DeviceArray.from_host(u)
        d_f = H.DeviceArray.zeros((n,))
        energies.append(b.pair_step_device(d_r, d_f, n, d_s, d_u, ns, est=ns))
        forces.append(d_f.to_host())
        states.append([b.get(k) for k in ("cum_bias", "overflow_left", "overflow_right", "b_skip_hill_add", "hills_added")])
    v, dv = b.gauss.download()
    out = dict(v=v, dv=dv, hist=np.array(b.hist.values), e=np.array(energies), f=np.array(forces), st=np.array(states),
               fused=b.get("fused_steps"), redos=b.get("bound_redos"), log=open(str(workdir / ("HILLS_%s_0" % tag))).read())
    del b
    return out


@pytest.mark.parametrize("mode", [1])
@pytest.mark.parametrize("name", list(CASES))
def test_one_launch_equals_two_launches(name, mode, workdir):
    case = CASES[name]
    one = run("one", case, workdir, mode)
    two = run("two", case, workdir, 0)
    # (a step that finds the overflow buffer still filled after its flush adds no new hills: edm_bias.cpp:534-535)
    want = 1 if name == "limiter_binds" else case["steps"] - 1
    assert one["fused"] >= want and two["fused"] == 0, (one["fused"], two["fused"])
    for k in ("v", "dv", "hist", "e", "f", "st"):
        assert np.array_equal(one[k], two[k]), k
    assert one["log"] == two["log"]
    assert one["v"].max() > 0 and np.abs(one["f"][-1]).max() > 0
    if name == "limiter_binds":
        assert one["st"][:, 2].max() > 0          # the overflow buffer was used
    if name == "below_limit":
        assert one["st"][:, 2].max() == 0
    if name == "slots_overflow":
        assert one["redos"] >= 1 and two["redos"] == 0


@pytest.mark.parametrize("mode", [1])
def test_one_launch_against_oracle(mode, workdir, oracle_lib):
    """The one-launch step against the oracle executing the reference's per-pair loop (pre_add_hill, update_force per
    pair, add_hill per sample, post_add_hill)."""
    text = BASE + "hill_density 50\nbias_per_step 0.3\nbias_spacing 0.001\nbias_sigma 0.05\n"
    cfgs = {}
    for tag in ("gpu", "ora"):
        cfgs[tag] = str(workdir / (tag + ".edm"))
        open(cfgs[tag], "w").write(text + "hills_filename %s/H_%s\nhistogram_filename %s/HIST_%s\n" % (workdir, tag, workdir, tag))
    b = H.Bias(cfgs["gpu"])
    b.set("debug_pair_step_mode", mode)
    ob = B.Bias(oracle_lib, cfgs["ora"])
    for x in (b, ob):
        x.setup(1.0, 1.0)
        x.subdivide([0], [2.8], [0], [2.8], [0], [0.3])
    n = 60_000
    for step in range(4):
        r = W.pair_distances(n, 5100 + step)
        u = W.uniform(5200 + step, n)
        d_r = H.DeviceArray.from_host(r)
        d_u = H.DeviceArray.from_host(u)
        d_f = H.DeviceArray.zeros((n,))
        e = b.pair_step_device(d_r, d_f, n, d_r, d_u, n, est=n)
        f = d_f.to_host()
        # the reference's order inside fix edm_pair: pre_add_hill, forces on the grid as it stands, then the hills
        ob.pre_add_hill(n)
        eo, fo = 0.0, np.zeros(n)
        for i in range(n):
            ei, fi = ob.update_force([r[i]])
            eo += ei
            fo[i] = fi[0]
        for i in range(n):
            ob.add_hill([r[i]], u[i])
        ob.post_add_hill()
        assert abs(e - eo) <= 1e-9 * max(1.0, abs(eo)), step
        assert np.allclose(f, fo, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(fo).max())), step
        assert abs(b.get("cum_bias") - ob.get("cum_bias")) <= 1e-10 * max(1.0, abs(ob.get("cum_bias")))
        keys = ("overflow_left", "overflow_right", "b_skip_hill_add", "hills_added")
        assert [b.get(k) for k in keys] == [ob.get(k) for k in keys], step
    assert b.get("fused_steps") >= 2
    v, dv = b.gauss.download()
    ogg = ob.gauss.grid
    assert np.allclose(v, ogg.values, rtol=1e-9, atol=1e-12 * np.abs(ogg.values).max())
    assert np.array_equal(b.hist.values, ob.hist.values)
