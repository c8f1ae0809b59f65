"""One rank of tests/test_gpu_two_ranks.py: drives the REAL multi-rank exchange code of the controller (packed
exchange, its overflow fallback, the synchronous record exchange, the sharded dense application with its all-gather
of integral slices and all-reduce of delta grids) with the host-staged carrier, several ranks sharing the one GPU
of the box.   usage: two_rank_worker.py <scenario> <rank> <nranks> <shm-name> <outdir>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import edm_amd.hip as H  # noqa: E402

import two_rank_cases as TC  # noqa: E402


def main():
    scenario, rank, nranks, shm, outdir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    H.require_gpu()
    case = TC.CASES[scenario]
    cfg = os.path.join(outdir, "rank%d.edm" % rank)
    with open(cfg, "w") as fh:
        fh.write(case["cfg"] + "hills_filename %s/HILLS_multi\nhistogram_filename %s/HIST_multi_%d\n" % (outdir, outdir, rank))
    b = H.Bias(cfg)
    b.comm_init_shm(shm, nranks, rank)
    out = TC.drive(H, b, case, rank, nranks)
    b.write_bias(os.path.join(outdir, "BIAS_multi"), 1)    # rank 0 writes, the others only wait (single writer)
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **out)
    del b
    print("rank %d done" % rank)


if __name__ == "__main__":
    main()
