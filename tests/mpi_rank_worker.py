"""One rank of tests/test_gpu_two_ranks.py::test_two_ranks_vs_reference_mpi_build: the scenarios of tests/mpi_cases.py
(which the REAL reference's MPI build ran under mpiexec -n 2, oracle/gen_golden_mpi.py) through the HIP controller's
multi-rank exchange with the host-staged carrier.   usage: mpi_rank_worker.py <case> <rank> <nranks> <shm-name> <outdir>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import edm_amd.hip as H  # noqa: E402

import mpi_cases as MC  # noqa: E402


def main():
    name, rank, nranks, shm, outdir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    H.require_gpu()
    spec = MC.MPI_CASES[name]
    cfg = os.path.join(outdir, "rank%d.edm" % rank)
    with open(cfg, "w") as fh:
        fh.write(spec["cfg"] + "\nhills_filename %s/HILLS_mpi\nhistogram_filename %s/HIST_mpi_%d\n" % (outdir, outdir, rank))
    b = H.Bias(cfg)
    b.comm_init_shm(shm, nranks, rank)
    b.setup(1.0, 1.0)
    b.subdivide([spec["lo"]], [spec["hi"]], [spec["lo"]], [spec["hi"]], [0], [spec["skin"]])
    cum, ovf, hadd = [], [], []
    E, F = [], []
    last_calls = spec.get("nmax", 0)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pairfix_cases as PF

    for step in range(spec["steps"]):
        if spec["mode"] == "pair_loop":
            # the rewritten fix's hill step in the reference's order (edm_hip_bias_pair_step_ordered_host)
            r, second, ru = MC.pair_loop_inputs(name, step, rank)
            xs, us = PF.staged_samples(r, second, ru)
            f = np.zeros(len(r))
            E.append(b.pair_step_ordered_host(r, f, PF.first_calls(second), xs, us, est=last_calls))
            F.append(f)
            last_calls = len(xs)
        elif spec["mode"] == "explicit":
            b.pre_add_hill(1)
            b.add_hill(spec["hills"][rank], 1.0)
            b.post_add_hill()
        else:
            pos, ru = MC.mpi_inputs(name, step, rank)
            b.add_hills(pos, ru, -1)
        cum.append(b.get("cum_bias"))
        ovf.append([int(b.get("overflow_left")), int(b.get("overflow_right")), int(b.get("b_skip_hill_add"))])
        hadd.append(int(b.get("hills_added")))
    v, dv = b.gauss.download()
    b.write_bias(os.path.join(outdir, "BIAS_mpi"), 0)    # the MPI build's write_bias = multi_write; rank 0 writes
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), values=v, derivs=dv, hist=b.hist.values, cum_bias=np.array(cum),
             overflow=np.array(ovf, dtype=np.int64), hills_added=np.array(hadd, dtype=np.int64),
             total_volume=b.get("total_volume"), hill_density=b.get("hill_density"), hill_prefactor=b.get("hill_prefactor"),
             energy=np.array(E), force=np.array(F))
    del b
    print("rank %d done" % rank)


if __name__ == "__main__":
    main()
