"""N > 1 path on CPU: world_size-2 gloo run of the hill-exchange protocol (edm_amd.parallel, the
host mirror of csrc/edm_bias.cpp:exchange_hills) with the CPU oracle doing the arithmetic."""
import os
import subprocess
import sys

import numpy as np

import edm_amd.workloads as W

from conftest import ROOT


def test_shard_bounds_partition():
    for n in (0, 1, 7, 1 << 20, 38_800_000):
        for world in (1, 2, 3, 8):
            edges = [W.shard_bounds(n, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
            sizes = [e[1] - e[0] for e in edges]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_replicated_replay(tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29613", os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    # every rank replays the same rank-major list: replicated state is bit-identical
    for key in ("grid", "hist", "cum", "overflow"):
        assert np.array_equal(r0[key], r1[key]), key
    assert r0["grid"].max() > 0 and r0["hist"].sum() > 0
    # cum_bias counts each hill once per rank (edm_bias.cpp:925), compensated by total_volume
    assert np.all(np.diff(r0["cum"]) > 0)


def test_two_rank_packed_exchange_and_sharded_dense(tmp_path, oracle_lib):
    """(i) the fixed-size packet exchange of a stochastic step yields the same rank-major list as the padded
    two-collective exchange, and an overflowing packet makes every rank fall back; (ii) the sharded dense
    application (own slice -> delta grid; integrals and delta grid all-reduced) equals one process applying
    every hill, to rounding."""
    from oracle import binding as B

    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29614", os.path.join(ROOT, "tests", "dist_worker2.py"), str(tmp_path)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    r0 = np.load(tmp_path / "w2_rank0.npz")
    r1 = np.load(tmp_path / "w2_rank1.npz")
    for key in ("merged", "added", "delta"):
        assert np.array_equal(r0[key], r1[key]), key      # all ranks hold identical results
    # the global list is the accepted samples in sample order (rank-major == shard order)
    r_all = W.pair_distances(30000, 41)
    u_all = W.uniform(42, 30000)
    assert np.array_equal(r0["merged"][:, 0], r_all[u_all < 0.004])
    # one process applying all hills
    hills = W.pair_distances(6000, 43).reshape(-1, 1)
    g = B.Gauss.create(oracle_lib, [0.0], [2.8], [0.001], [0], 1, [0.05])
    ref_added = np.array([g.add_value(x, 2e-4) for x in hills])
    assert np.array_equal(r0["added"], ref_added)         # (each integral is computed by exactly one rank)
    assert np.allclose(r0["delta"], g.grid.values, rtol=1e-12, atol=1e-15)
