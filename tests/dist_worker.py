"""Worker for tests/test_distributed_cpu.py: world_size ranks over gloo on CPU.

Each rank owns a contiguous shard of the step's samples, accepts hills with its own uniforms,
exchanges the records with edm_amd.parallel.merge_rank_major and replays the rank-major global
list on the CPU oracle (test infrastructure).  Rank r writes its final state to <out>/rank<r>.npz.
"""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import edm_amd.parallel as P  # noqa: E402
import edm_amd.workloads as W  # noqa: E402
from oracle import binding as B  # noqa: E402

CFG = ("tempering 0\nhill_prefactor 0.5\nhill_density 40\nbias_per_step 0.3\ndimension 1\nbox_low 0\nbox_high 2.8\n"
       "bias_spacing 0.001\nbias_sigma 0.05\n")


def run(out, world_override=None):
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    # RCCL bootstrap stand-in: rank 0 creates an id, everyone receives it
    ident = [bytes(range(128)) if rank == 0 else None]
    dist.broadcast_object_list(ident, src=0)
    assert ident[0] == bytes(range(128))
    cfg = os.path.join(out, "r%d.edm" % rank)
    open(cfg, "w").write(CFG + "hills_filename %s/HILLS\nhistogram_filename %s/HIST%d\n" % (out, out, rank))
    lib = B.load("oracle")
    b = B.Bias(lib, cfg)
    b.setup(1.0, 1.0)
    b.subdivide([0], [2.8], [0], [2.8], [0], [0.3])
    # EDMBias::subdivide under MPI (edm_bias.cpp:175-180): per-system density and prefactor
    b.set("hill_density", b.get("hill_density") / world)
    b.set("hill_prefactor", b.get("hill_prefactor") / world)
    b.set("total_volume", b.get("total_volume") * world)
    n_total = 40000
    cum = []
    for step in range(4):
        r_all = W.pair_distances(n_total, 700 + step)
        u_all = W.uniform(800 + step, n_total)
        lo, hi = P.shard_bounds(n_total, world, rank)
        r, u = r_all[lo:hi], u_all[lo:hi]
        est = 2 * (hi - lo)
        thr = b.get("hill_density") / est
        mine = r[u < thr].reshape(-1, 1)
        merged, counts = P.merge_rank_major(dist, mine)
        assert sum(counts) == merged.shape[0]
        b.pre_add_hill(est)
        for x in merged:
            b.add_hill(x, 0.0)  # already accepted: replay like a received hill (u = 0 < thr)
        b.post_add_hill()
        # update_height sums every rank's temp_hill_cum_: N-fold counting, as in the reference
        step_bias = b.get("cum_bias") - (cum[-1] if cum else 0.0)
        cum.append((cum[-1] if cum else 0.0) + P.replicated_totals(dist, step_bias))
        b.set("cum_bias", cum[-1])
    np.savez(os.path.join(out, "rank%d.npz" % rank), grid=b.gauss.grid.values, hist=b.hist.values, cum=np.array(cum),
             overflow=np.array([b.get("overflow_left"), b.get("overflow_right")]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    run(sys.argv[1])
