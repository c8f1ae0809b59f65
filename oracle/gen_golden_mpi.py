#!/usr/bin/env python3
"""Generates tests/golden/mpi2_* by running the REAL reference's MPI build
(oracle/_ref/libedm_ref_mpi.so = /root/reference/lib/*.cpp compiled WITHOUT -DEDM_SERIAL by `make -C oracle ref_mpi`)
under `mpiexec -n 2`, one python process per rank, each driving its EDMBias through oracle/ref_shim.cpp.
Fixtures hold inputs and expected outputs only -- numbers and the text files the reference writes.

Run in the build container (where /root/reference and an MPI launcher exist):
    python oracle/gen_golden_mpi.py            # spawns: mpiexec -n 2 python oracle/gen_golden_mpi.py --worker
TEST INFRASTRUCTURE ONLY.
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GOLDEN = os.path.join(ROOT, "tests", "golden")
MPIEXEC = os.environ.get("MPIEXEC", "/opt/conda/bin/mpiexec")


def worker(tmp):
    from oracle import binding as B
    import mpi_cases as MC

    lib = B.load("ref_mpi")
    assert lib.fn("is_mpi_build")() == 1
    rank, size = lib.fn("mpi_rank")(), lib.fn("mpi_size")()
    assert size == MC.NRANKS
    for name in sorted(MC.MPI_CASES):
        spec = MC.MPI_CASES[name]
        cfg = os.path.join(tmp, "%s_%d.edm" % (name, rank))
        hills = os.path.join(tmp, "HILLS_" + name)
        with open(cfg, "w") as fh:
            fh.write(spec["cfg"] + "\nhills_filename %s\nhistogram_filename %s.hist\n" % (hills, hills))
        b = B.Bias(lib, cfg)
        b.setup(1.0, 1.0)
        # the replicated decomposition of fix edm_pair (fix_edm_pair.cpp:95-104): every rank passes the whole range
        b.subdivide([spec["lo"]], [spec["hi"]], [spec["lo"]], [spec["hi"]], [0], [spec["skin"]])
        rec = dict(cum=[], temp_before_post=[], overflow=[], hills_added=[])
        E, F, NC = [], [], []
        last_calls = spec.get("nmax", 0)
        for step in range(spec["steps"]):
            if spec["mode"] == "pair_loop":
                r, second, ru = MC.pair_loop_inputs(name, step, rank)
                e, f, nc = b.pair_loop(r, second, ru, 1, last_calls)
                last_calls = nc
                E.append(e)
                F.append(f)
                NC.append(nc)
            elif spec["mode"] == "explicit":
                b.pre_add_hill(1)
                b.add_hill(spec["hills"][rank], 1.0)
                b.post_add_hill()
            else:
                pos, ru = MC.mpi_inputs(name, step, rank)
                b.add_hills(pos, ru, -1)
            rec["cum"].append(b.get("cum_bias"))
            rec["overflow"].append([int(b.get("overflow_left")), int(b.get("overflow_right")), int(b.get("b_skip_hill_add"))])
            rec["hills_added"].append(int(b.get("hills_added")))
        gg = b.gauss.grid
        np.savez_compressed(os.path.join(GOLDEN, "mpi2_%s_rank%d.npz" % (name, rank)),
                            grid_values=gg.values.copy(), grid_derivs=gg.derivs.copy(), hist=b.hist.values.copy(),
                            cum_bias=np.array(rec["cum"]), overflow=np.array(rec["overflow"], dtype=np.int64),
                            hills_added=np.array(rec["hills_added"], dtype=np.int64),
                            total_volume=b.get("total_volume"), hill_density=b.get("hill_density"),
                            hill_prefactor=b.get("hill_prefactor"), mpi_neighbor_count=b.get("mpi_neighbor_count"),
                            energy=np.array(E), force=np.array(F), ncalls=np.array(NC, dtype=np.int64))
        # write_bias of the MPI build is the collective multi_write (edm_bias.cpp:224-235, grid.h:509-674): rank 0's file
        bias_file = os.path.join(tmp, "BIAS_" + name)
        b.write_bias(bias_file)
        lib.fn("mpi_barrier")()
        del b
        shutil.copy("%s_%d" % (hills, rank), os.path.join(GOLDEN, "mpi2_%s_rank%d.hills.txt" % (name, rank)))
        if rank == 0 and name == "pair_density":
            shutil.copy(bias_file, os.path.join(GOLDEN, "mpi2_%s.multiwrite.grid" % name))
        lib.fn("mpi_barrier")()
    lib.fn("mpi_finalize")()


def main():
    if "--worker" in sys.argv:
        worker(sys.argv[sys.argv.index("--worker") + 1])
        return
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref_mpi"], stdout=subprocess.DEVNULL)
    tmp = tempfile.mkdtemp(prefix="edm_golden_mpi_")
    env = dict(os.environ)
    subprocess.check_call([MPIEXEC, "-n", "2", sys.executable, os.path.abspath(__file__), "--worker", tmp], env=env)
    shutil.rmtree(tmp)
    print("MPI golden fixtures written to", GOLDEN)


if __name__ == "__main__":
    main()
