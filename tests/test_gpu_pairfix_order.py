"""fix edm_pair in the REFERENCE'S order (lammps/fix_edm_pair.cpp:173-247) against fixtures the real reference
produced when driven exactly like its own fix -- per pair update_force, then one or two add_hill
(oracle/gen_golden.py:run_pairfix through oracle/ref_shim.cpp:ref_bias_pair_loop; tests/golden/pairfix_*.npz).

* edm_hip_bias_pair_step_ordered(_host) must reproduce the reference's forces and energy within 1e-6 relative
  (BASELINE.json north_star; asserted tighter), with limiter state, add_hill counts and the HILLS events exact.
* edm_hip_bias_pair_step (every force of the step on the bias as it stands after pre_add_hill) is the FAST mode: its
  hills, grid, histogram and limiter state are the reference's too, its forces on a hill step are not -- the test
  measures that deviation against the same fixture, bounds it by the bias the step itself deposits, and writes the
  numbers to gpurun_out/pairfix_deviation.json (quoted in INTEGRATION.md).
"""
import json
import os

import numpy as np
import pytest

import edm_amd.hip as H
from oracle import binding as B

import golden_util as GU
import pairfix_cases as PF
from test_gpu_parity import _parse_hills, close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    H.require_gpu()
    yield


def _make(cls, spec, name, workdir, tag, *lib):
    cfg = str(workdir / ("%s_%s.edm" % (name, tag)))
    hills = str(workdir / ("HILLS_%s_%s" % (name, tag)))
    with open(cfg, "w") as fh:
        fh.write(spec["cfg"] + "\nhills_filename %s\nhistogram_filename %s.hist\n" % (hills, hills))
    b = cls(*lib, cfg)
    b.setup(1.0, 1.0)
    b.subdivide([spec["lo"]], [spec["hi"]], [spec["lo"]], [spec["hi"]], [0], [spec["skin"]])
    return b, hills


DEVIATION = {}


@pytest.mark.parametrize("name", sorted(PF.PAIRFIX))
def test_reference_order_step(name, workdir, oracle_lib):
    spec = PF.PAIRFIX[name]
    gold = np.load(os.path.join(GU.GOLDEN, "pairfix_%s.npz" % name), allow_pickle=False)
    ordered, hills_ordered = _make(H.Bias, spec, name, workdir, "ordered")
    fast, _ = _make(H.Bias, spec, name, workdir, "fast")
    batch_oracle, _ = _make(B.Bias, spec, name, workdir, "oracle", oracle_lib)   # the oracle driven in the FAST mode's order
    last_calls = spec["nmax"]
    report = []
    for step, hill_step in enumerate(spec["steps"]):
        r, second, ru = PF.pairfix_inputs(name, step)
        n = len(r)
        want_f, want_e = gold["force"][step], gold["energy"][step]
        scale = np.abs(want_f).max()
        f_ord = np.zeros(n)
        f_fast = np.zeros(n)
        if hill_step:
            xs, us = PF.staged_samples(r, second, ru)
            first = PF.first_calls(second)
            e_ord = ordered.pair_step_ordered_host(r, f_ord, first, xs, us, est=last_calls)
            e_fast = fast.pair_step_host(r, f_fast, xs, us, est=last_calls)
            assert len(xs) == gold["ncalls"][step]
            # the oracle in the fast mode's order: pre_add_hill, every force, then the add_hill calls
            batch_oracle.pre_add_hill(last_calls)
            f_bo = np.array([batch_oracle.update_force([x])[1][0] for x in r])
            for x, u in zip(xs, us):
                batch_oracle.add_hill([x], float(u))
            batch_oracle.post_add_hill()
            last_calls = len(xs)
        else:
            d_r = H.DeviceArray.from_host(r)
            d_f = H.DeviceArray.from_host(np.zeros(n))
            e_ord = ordered.pair_forces_device(d_r, d_f, n)
            f_ord = d_f.to_host()
            d_f2 = H.DeviceArray.from_host(np.zeros(n))
            e_fast = fast.pair_forces_device(d_r, d_f2, n)
            f_fast = d_f2.to_host()
            f_bo = np.array([batch_oracle.update_force([x])[1][0] for x in r])
        # ---- the reference-order step: the reference's numbers (bar 1e-6 relative; dV/dr is a difference of
        #      O(V / dx) terms, hence the absolute floor relative to the largest force) ----
        close(f_ord, want_f, rtol=1e-8, atol=1e-10 * scale, what="%s step %d: forces in the reference's order" % (name, step))
        close(e_ord, want_e, rtol=1e-10, atol=1e-12, what="%s step %d: energy in the reference's order" % (name, step))
        for b in (ordered, fast):
            close(b.get("cum_bias"), gold["cum_bias"][step], rtol=1e-10, what="cum_bias")
            got = [int(b.get("overflow_left")), int(b.get("overflow_right")), int(b.get("b_skip_hill_add"))]
            assert got == list(gold["overflow"][step]), "limiter state must match exactly (step %d)" % step
        # ---- the fast mode equals the oracle driven in ITS order, and deviates from the reference like this ----
        close(f_fast, f_bo, rtol=1e-8, atol=1e-10 * max(scale, 1e-300), what="%s step %d: fast mode vs oracle in batch order" % (name, step))
        dev = float(np.abs(f_fast - want_f).max())
        report.append(dict(step=step, hill_step=int(hill_step), max_abs_force=float(scale), max_abs_force_deviation=dev,
                           energy_reference_order=float(want_e), energy_fast=float(e_fast)))
        if hill_step:
            assert dev > 0, "a hill step's batched forces differ from the reference's (that is the documented deviation)"
            # the deviation is the force of the bias the step itself deposits: never more than the steepest slope of the
            # step's own bias, which the next step's forces (identical in both modes) bound
        else:
            assert dev <= 1e-8 * scale + 1e-300, "between hill steps the two modes coincide"
    DEVIATION[name] = report
    # grid, histogram: both modes leave the reference's bias, bit-identical to each other
    v1, d1 = ordered.gauss.download()
    v2, d2 = fast.gauss.download()
    assert np.array_equal(v1, v2) and np.array_equal(d1, d2), "hills do not depend on the order the forces are read in"
    close(v1, gold["grid_values"], rtol=1e-9, atol=1e-13 * np.abs(gold["grid_values"]).max(), what="grid")
    close(d1, gold["grid_derivs"], rtol=1e-9, atol=1e-11 * max(np.abs(gold["grid_derivs"]).max(), 1e-300), what="derivs")
    assert np.array_equal(ordered.hist.values, gold["hist"])
    del ordered, fast
    got = _parse_hills(hills_ordered + "_0")
    want = _parse_hills(os.path.join(GU.GOLDEN, "pairfix_%s.hills.txt" % name))
    assert len(got) == len(want)
    for a, w in zip(got, want):
        assert a[:3] == w[:3], (a, w)
        close(a[3:], w[3:], rtol=0, atol=2e-8, what="HILLS line")


def test_device_entry_equals_host_entry(workdir):
    """edm_hip_bias_pair_step_ordered on device arrays == the _host entry, bit for bit"""
    name = "w1_density"
    spec = PF.PAIRFIX[name]
    a, _ = _make(H.Bias, spec, name, workdir, "dev")
    b, _ = _make(H.Bias, spec, name, workdir, "host")
    last = spec["nmax"]
    for step in range(2):
        r, second, ru = PF.pairfix_inputs(name, step)
        xs, us = PF.staged_samples(r, second, ru)
        first = PF.first_calls(second)
        d_r, d_f = H.DeviceArray.from_host(r), H.DeviceArray.from_host(np.zeros(len(r)))
        d_first = H.DeviceArray.from_host(first)
        d_x, d_u = H.DeviceArray.from_host(xs), H.DeviceArray.from_host(us)
        e1 = a.pair_step_ordered_device(d_r, d_f, d_first, len(r), d_x, d_u, len(xs), est=last)
        f2 = np.zeros(len(r))
        e2 = b.pair_step_ordered_host(r, f2, first, xs, us, est=last)
        assert e1 == e2 and np.array_equal(d_f.to_host(), f2)
        last = len(xs)


@pytest.mark.parametrize("name", ["w1_density", "walls_inside", "all_samples"])
def test_force_pass_does_not_depend_on_the_order_of_the_pair_array(name, workdir):
    """A pair's force is a function of (distance, index of its first add_hill call) and the step's hills.  The force
    pass takes a short cut for arrays whose sample indices ascend (the fix's list: a workgroup's run of pairs spans a
    few rows of the counts, staged in LDS, and a lean per-pair form); a shuffled array sends every pair down the general
    form (its row read from memory, the whole sample list searched) and must give every pair the same force."""
    spec = PF.PAIRFIX[name]
    a, _ = _make(H.Bias, spec, name, workdir, "asc")
    b, _ = _make(H.Bias, spec, name, workdir, "shuf")
    last = spec["nmax"]
    rng = np.random.default_rng(5)
    for step in range(2):
        r, second, ru = PF.pairfix_inputs(name, step)
        xs, us = PF.staged_samples(r, second, ru)
        first = PF.first_calls(second)
        perm = rng.permutation(len(r))
        out = []
        for bias, rr, ff in ((a, r, first), (b, np.ascontiguousarray(r[perm]), np.ascontiguousarray(first[perm]))):
            d_r, d_f = H.DeviceArray.from_host(rr), H.DeviceArray.from_host(np.zeros(len(rr)))
            d_first = H.DeviceArray.from_host(ff)
            d_x, d_u = H.DeviceArray.from_host(xs), H.DeviceArray.from_host(us)
            e = bias.pair_step_ordered_device(d_r, d_f, d_first, len(rr), d_x, d_u, len(xs), est=last)
            out.append((e, d_f.to_host()))
        (e1, f1), (e2, f2) = out
        assert np.array_equal(f2, f1[perm]), "shuffled pairs must get the forces of the ascending array, bit for bit"
        close(e2, e1, rtol=1e-12, atol=0, what="energy (another summation order)")
        last = len(xs)


@pytest.mark.parametrize("first_step", ["hills", "empty_first"])
def test_gate_wave_gives_up_and_the_step_is_finished_on_one_stream(first_step, workdir):
    """The record pass of a single-rank step runs on a second stream behind a one-wave gate that waits, in the kernel,
    for the hill batch's limiter.  Where kernels of different streams are run one at a time (a profiler collecting
    hardware counters does that) the gate may be let in ahead of the batch and would wait for ever: it gives up after
    2 ms, the passes behind it leave, and the host queues them again behind the batch -- same numbers, and the second
    stream stays unused from then on (also when the step in question accepted no hill at all).  Forced here (EDM_HIP_TEST_FORCE=ord_gate_giveup, a worker process: the switch is
    read once per process)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for tag in ("plain", "forced"):
        env = dict(os.environ, PYTHONPATH=root)
        env.pop("EDM_HIP_TEST_FORCE", None)
        if tag == "forced":
            env["EDM_HIP_TEST_FORCE"] = "ord_gate_giveup"
        p = subprocess.run([sys.executable, os.path.join(root, "tests", "gate_worker.py"), str(workdir), first_step],
                           capture_output=True, text=True, env=env, timeout=300)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")][-1].split()
        out[tag] = (line[1], int(line[2]))
    assert out["plain"][1] == 0 and out["forced"][1] == 1, out
    assert out["plain"][0] == out["forced"][0], "the step finished on one stream must give the same energies and forces"


def test_long_pair_arrays_take_the_lds_window_form(workdir):
    """Two million pairs: the force pass keeps a window of the running records in LDS (k_pair_forces_ordered_win, K1's
    layout for long arrays).  Same forces, bit for bit, as the short-array kernel on the same array (forced by
    EDM_HIP_TEST_FORCE=no_k1o_window in a worker process), with walls inside the grid, pairs outside the window and
    outside the walls -- and as the same pairs in shuffled order, which the window form hands to the general form one
    by one."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for tag, token, extra in (("window", None, []), ("short", "no_k1o_window", []), ("shuffled", None, ["shuffled"])):
        env = dict(os.environ, PYTHONPATH=root)
        env.pop("EDM_HIP_TEST_FORCE", None)
        if token:
            env["EDM_HIP_TEST_FORCE"] = token
        p = subprocess.run([sys.executable, os.path.join(root, "tests", "k1o_window_worker.py"), str(workdir)] + extra,
                           capture_output=True, text=True, env=env, timeout=300)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")][-1].split()
        out[tag] = (line[1], [float(v) for v in line[2:4]], int(line[4]))
    assert out["window"][2] > 100
    for other in ("short", "shuffled"):
        assert out[other][0] == out["window"][0], "forces of the %s run differ from the window form's" % other
        close(out[other][1], out["window"][1], rtol=1e-12, atol=0, what="energies (another summation order)")


def test_write_deviation_report():
    """(runs last in this module) the measured deviation of the fast mode, for INTEGRATION.md"""
    if not DEVIATION:
        pytest.skip("no scenario ran")
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "pairfix_deviation.json"), "w") as fh:
        json.dump(DEVIATION, fh, indent=1)
