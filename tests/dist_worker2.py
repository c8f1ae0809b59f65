"""Worker for tests/test_distributed_cpu.py (second scenario): the packed stochastic exchange and the sharded
dense application, world_size ranks over gloo on CPU, oracle arithmetic (test infrastructure)."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import edm_amd.parallel as P  # noqa: E402
import edm_amd.workloads as W  # noqa: E402
from oracle import binding as B  # noqa: E402


def run(out):
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    lib = B.load("oracle")
    res = {}
    # ---- packed exchange == padded two-collective exchange; overflow poisons every rank alike ----
    n_total = 30000
    r_all = W.pair_distances(n_total, 41)
    u_all = W.uniform(42, n_total)
    lo, hi = P.shard_bounds(n_total, world, rank)
    mine = r_all[lo:hi][u_all[lo:hi] < 0.004].reshape(-1, 1)
    a, ca = P.merge_rank_major(dist, mine)
    b, cb = P.merge_packets(dist, mine, bound=256)
    assert ca == cb and np.array_equal(a, b)
    none, cc = P.merge_packets(dist, mine, bound=max(1, min(ca) - 1) if min(ca) > 1 else 1)
    assert none is None and cc == ca            # (some rank overflows -> every rank falls back)
    res["merged"] = b
    # ---- sharded dense application: own slice -> delta grid, integrals and delta all-reduced ----
    hills = W.pair_distances(6000, 43).reshape(-1, 1)
    h = 2e-4
    own_lo, own_hi = P.shard_bounds(len(hills), world, rank)
    g = B.Gauss.create(lib, [0.0], [2.8], [0.001], [0], 1, [0.05])
    added = np.zeros(len(hills))
    for i in range(own_lo, own_hi):
        added[i] = g.add_value(hills[i], h)
    added = P.allreduce_sum(dist, added)                      # every rank now has every hill's integral
    delta = P.allreduce_sum(dist, g.grid.values)              # ... and the sum of the ranks' delta grids
    res["added"] = added
    res["delta"] = delta
    np.savez(os.path.join(out, "w2_rank%d.npz" % rank), **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    run(sys.argv[1])
