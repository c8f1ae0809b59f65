"""Loaders for tests/golden (outputs of the real reference; see oracle/gen_golden.py)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXTURES = os.path.join(GOLDEN, "ref_fixtures")


def gauss_scenarios():
    with open(os.path.join(GOLDEN, "gauss_scenarios.json")) as fh:
        return json.load(fh)


def gauss_data(name):
    return np.load(os.path.join(GOLDEN, "gauss_%s.npz" % name), allow_pickle=False)


def controller_cases():
    with open(os.path.join(GOLDEN, "controller.json")) as fh:
        return json.load(fh)


def controller_data(name):
    return np.load(os.path.join(GOLDEN, "ctrl_%s.npz" % name), allow_pickle=False)


def kats():
    with open(os.path.join(GOLDEN, "kats.json")) as fh:
        return json.load(fh)


def controller_inputs(case, step, dim, lo, hi):
    """Regenerates the seeded inputs of oracle/gen_golden.py:ctrl_inputs."""
    import edm_amd.workloads as W

    names = sorted(c["name"] for c in controller_cases())
    seed = 5000 + 100 * names.index(case["name"]) + step
    n = case["n"]
    lo = np.asarray(lo, dtype=float)
    hi = np.asarray(hi, dtype=float)
    pos = np.zeros((n, 3))
    pos[:, :dim] = lo + W.uniform(seed, n * dim).reshape(n, dim) * (hi - lo) * 1.04 - 0.02 * (hi - lo)
    ru = W.uniform(seed + 50, n)
    mask = (W.splitmix64(seed + 77, n) % np.uint64(4)).astype(np.int32)
    return pos, ru, mask
