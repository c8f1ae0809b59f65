"""Scenarios of the reference's MPI build (lib/ compiled WITHOUT -DEDM_SERIAL) run under `mpiexec -n 2`
(oracle/gen_golden_mpi.py -> tests/golden/mpi2_*): the replicated pair-CV decomposition of fix edm_pair, where every
rank's grid spans the whole r range and the ranks broadcast their hills to each other (edm_bias.cpp:630-706).  Shared
by the generator and the tests, which regenerate the seeded per-rank inputs here.  Pure numpy: inputs only.
"""
import numpy as np

import edm_amd.workloads as W

NRANKS = 2

MPI_CASES = {
    # one explicit hill per rank on the notebook's 1-D grid, limit not binding: the known answer of SURVEY 8c(iv)
    # (cum_bias counts every hill once per rank, total_volume is summed over the ranks)
    "two_hills": dict(
        cfg="tempering 0\nhill_prefactor 1\nbias_per_step 100\ndimension 1\nbox_low 0\nbox_high 10\n"
            "bias_spacing 0.01\nbias_sigma 0.5",
        lo=0.0, hi=10.0, skin=0.0, mode="explicit", steps=1, hills=[[2.5], [7.25]]),
    # stochastic replicated pair-CV steps, limit not binding (hill_density and hill_prefactor are divided by the
    # rank count in subdivide, edm_bias.cpp:175-180)
    "pair_density": dict(
        cfg="tempering 0\nhill_prefactor 0.5\nhill_density 40\nbias_per_step 1000\ndimension 1\nbox_low 0\nbox_high 2.8\n"
            "bias_spacing 0.001\nbias_sigma 0.05",
        lo=0.0, hi=2.8, skin=0.3, mode="array", steps=3, n=2048),
    # the same with a binding limit: every rank limits against ITS OWN running sum in ITS OWN replay order
    # (own hills first, then the other ranks' in rank order), so the reference's replicas drift apart
    "pair_density_limit": dict(
        cfg="tempering 0\nhill_prefactor 0.5\nhill_density 40\nbias_per_step 0.3\ndimension 1\nbox_low 0\nbox_high 2.8\n"
            "bias_spacing 0.001\nbias_sigma 0.05",
        lo=0.0, hi=2.8, skin=0.3, mode="array", steps=3, n=2048),
    # the pair fix's own loop on every rank (lammps/fix_edm_pair.cpp:173-247): per pair update_force, then its add_hill
    # calls -- a rank's pairs see that rank's earlier hills of the step; the other ranks' arrive in post_add_hill
    "pair_loop": dict(
        cfg="tempering 0\nhill_prefactor 0.5\nhill_density 60\nbias_per_step 1000\ndimension 1\nbox_low 0\nbox_high 2.8\n"
            "bias_spacing 0.001\nbias_sigma 0.05",
        lo=0.0, hi=2.8, skin=0.3, mode="pair_loop", steps=3, n=3000, nmax=4000),
}


def pair_loop_inputs(name, step, rank):
    """(r[n], second[n], uniforms[2 n]) of `rank` at `step` (pair_loop cases)"""
    spec = MPI_CASES[name]
    seed = 14000 + 1000 * sorted(MPI_CASES).index(name) + 10 * step + rank
    n = spec["n"]
    r = np.cbrt(W.uniform(seed, n) * (2.8 ** 3 - 0.85 ** 3) + 0.85 ** 3)
    second = (W.uniform(seed + 3, n) < 0.7).astype(np.int32)
    return r, second, W.uniform(seed + 6, 2 * n)


def mpi_inputs(name, step, rank):
    """(positions [n, 3], uniforms [n]) of `rank` at `step` (array-mode cases)"""
    spec = MPI_CASES[name]
    seed = 12000 + 1000 * sorted(MPI_CASES).index(name) + 10 * step + rank
    n = spec["n"]
    pos = np.zeros((n, 3))
    pos[:, 0] = 0.7 + 2.1 * W.uniform(seed, n)
    return pos, W.uniform(seed + 5, n)
