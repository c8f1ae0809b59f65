"""The multi-rank hill exchange with MORE THAN ONE RANK: two processes share the box's GPU and run the controller's
real exchange code (csrc/edm_bias.cpp exchange_hills / packed exchange / overflow fallback, apply_hills' sharded dense
application) over the host-staged carrier (edm_hip_bias_comm_init_shm; the RCCL carrier needs one GPU per rank).
Replaces EDMBias::flush_buffers / update_height (edm_bias.cpp:630-706, :922-931).

Checks: every rank ends with the SAME bits (grid, histogram, limiter state, HILLS log), and they equal a single-process
controller fed the rank-major concatenation of the ranks' samples with the per-system density / prefactor the ranks
use after EDMBias::subdivide's split (edm_bias.cpp:175-180) -- bit for bit where the arithmetic is the same, at 1e-12
where the sharded application sums in a different order."""
import os
import subprocess
import sys
import uuid

import numpy as np
import pytest

import edm_amd.hip as H

import two_rank_cases as TC

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NR = 2


def _twin_cfg(case):
    """the single-process equivalent of NR ranks: density and prefactor as one rank holds them after the split"""
    cfg = case["cfg"]
    if "hill_density" in cfg:
        import re
        dens = float(re.search(r"hill_density (\S+)", cfg).group(1))
        pref = float(re.search(r"hill_prefactor (\S+)", cfg).group(1))
        cfg = cfg.replace("hill_density %g" % dens, "hill_density %.17g" % (dens / NR))
        cfg = cfg.replace("hill_prefactor %g" % pref, "hill_prefactor %.17g" % (pref / NR))
        if "bias_per_step" not in cfg:
            cfg += "bias_per_step %.17g\n" % pref
    return cfg


@pytest.mark.parametrize("scenario", sorted(TC.CASES))
def test_two_ranks_on_one_gpu(scenario, tmp_path):
    H.require_gpu()
    case = TC.CASES[scenario]
    shm = "/edm_test_%s" % uuid.uuid4().hex[:12]
    env = dict(os.environ, PYTHONPATH=ROOT)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "two_rank_worker.py"), scenario, str(r), str(NR), shm,
                               str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
             for r in range(NR)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, outs[r][-3000:])
    ranks = [np.load(str(tmp_path / ("rank%d.npz" % r))) for r in range(NR)]
    # 1. the replicas are identical, bit for bit
    for key in ("values", "derivs", "hist", "state"):
        assert np.array_equal(ranks[0][key], ranks[1][key]), "ranks differ in %s" % key
    assert ranks[0]["values"].max() > 0
    logs = [open(str(tmp_path / ("HILLS_multi_%d" % r))).read() for r in range(NR)]
    assert logs[0] == logs[1] and len(logs[0]) > 0
    # a single writer: rank 0 wrote the bias file, nobody truncated it afterwards
    assert os.path.getsize(str(tmp_path / "BIAS_multi")) > 1000
    # 2. the single-process twin
    cfg = str(tmp_path / "twin.edm")
    open(cfg, "w").write(_twin_cfg(case) + "hills_filename %s/HILLS_twin\nhistogram_filename %s/HIST_twin\n" % (tmp_path, tmp_path))
    b = H.Bias(cfg)
    twin = TC.drive(H, b, case, 0, NR, twin=True)
    del b
    dense = "hill_density" not in case["cfg"]
    st_m, st_t = ranks[0]["state"], twin["state"]
    assert np.array_equal(st_m[:, 1:], st_t[:, 1:]), "limiter state differs from the single-process twin"
    # cum_bias: every rank's step total is multiplied by the rank count (MPI_Allreduce of identical addends, :925)
    assert np.allclose(st_m[:, 0], NR * st_t[:, 0], rtol=1e-12 if dense else 0, atol=0)
    assert st_m[-1, 2] > 0 or st_m[:, 3].any() or dense, "the scenario should keep the limiter busy"
    if dense:
        vmax = np.abs(twin["values"]).max()
        assert np.allclose(ranks[0]["values"], twin["values"], rtol=1e-11, atol=1e-13 * vmax)
        assert np.allclose(ranks[0]["derivs"], twin["derivs"], rtol=1e-10, atol=1e-12 * np.abs(twin["derivs"]).max())
    elif case["all_accept_step"] >= 0:
        # the fallback applies its 1800 hills in hill groups whose number follows the EXPECTED batch size, which the
        # multi-rank redo does not have (it uses the actual count): same hills, same limiter decisions, the per-node
        # sums associated differently
        vmax = np.abs(twin["values"]).max()
        assert np.allclose(ranks[0]["values"], twin["values"], rtol=1e-12, atol=1e-14 * vmax)
        assert np.allclose(ranks[0]["derivs"], twin["derivs"], rtol=1e-11, atol=1e-13 * np.abs(twin["derivs"]).max())
        assert logs[0] == open(str(tmp_path / "HILLS_twin_0")).read()
    else:
        assert np.array_equal(ranks[0]["values"], twin["values"]) and np.array_equal(ranks[0]["derivs"], twin["derivs"])
        assert logs[0] == open(str(tmp_path / "HILLS_twin_0")).read()
    assert np.array_equal(ranks[0]["hist"], twin["hist"])
    if case["all_accept_step"] >= 0:
        assert ranks[0]["bound_redos"] == 1 and ranks[1]["bound_redos"] == 1, "the overflowing packets must send both ranks down the fallback"
    if "energies" in twin:
        # forces are evaluated per rank on the replicated grid: rank r's slice of the twin's force array
        n = case["n"]
        off = 0
        for r in range(NR):
            fr = ranks[r]["forces"].reshape(case["steps"], -1)
            ft = twin["forces"].reshape(case["steps"], -1)[:, off:off + fr.shape[1]]
            assert np.array_equal(fr, ft), "rank %d forces" % r
            off += fr.shape[1]
        assert np.allclose(ranks[0]["energies"] + ranks[1]["energies"], twin["energies"], rtol=1e-12)
