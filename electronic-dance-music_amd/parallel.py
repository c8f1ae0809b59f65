"""Host-side mirror of the multi-GPU hill exchange protocol (csrc/edm_bias.cpp:exchange_hills).

One process per GPU.  Per hill step every rank contributes the (position) records of the hills it
accepted; the records are all-gathered (counts first, then padded payloads) and concatenated in
RANK-MAJOR order, and every rank replays that same global list, so replicated bias grids stay
identical.  The C++ controller does this with two ncclAllGather calls on its HIP stream; this
module expresses the same protocol over ``torch.distributed`` (backend "nccl" = RCCL on GPUs,
"gloo" on CPUs) so the sharding arithmetic and ordering are testable without a GPU.
"""
import numpy as np

from .workloads import shard_bounds  # noqa: F401  (re-exported: contiguous sample shards)


def merge_rank_major(dist, local_records):
    """local_records: float64 [n_local, width] -> float64 [sum n_r, width], ranks in order."""
    import torch

    world = dist.get_world_size()
    local = np.ascontiguousarray(local_records, dtype=np.float64)
    width = local.shape[1] if local.ndim == 2 else 1
    local = local.reshape(-1, width)
    counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([local.shape[0]], dtype=torch.int64))
    counts = [int(c.item()) for c in counts]
    maxc = max(counts)
    if maxc == 0:
        return np.zeros((0, width)), counts
    send = torch.zeros(maxc, width, dtype=torch.float64)
    send[: local.shape[0]] = torch.from_numpy(local)
    recv = [torch.zeros(maxc, width, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(recv, send)
    merged = np.concatenate([recv[r][: counts[r]].numpy() for r in range(world)], axis=0)
    return merged, counts


def replicated_totals(dist, value):
    """update_height (edm_bias.cpp:922-931): sum of every rank's temp_hill_cum_."""
    import torch

    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t)
    return float(t.item())


def merge_packets(dist, local_positions, bound):
    """The packed exchange of a stochastic step (csrc/edm_bias.cpp:process_new_hills, k_select_prep /
    k_unpack_prep): every rank sends ONE fixed-size packet [count, x_0 .. x_{bound-1}] (bound positions of
    `dim` doubles, unused slots undefined), the packets are all-gathered, and the rank-major global list is
    cut out of them.  A rank whose count exceeds `bound` poisons the exchange: every rank returns None (and
    falls back to merge_rank_major), because all ranks see the same counts."""
    import torch

    world = dist.get_world_size()
    local = np.ascontiguousarray(local_positions, dtype=np.float64)
    dim = local.shape[1]
    packet = torch.zeros(1 + bound * dim, dtype=torch.float64)
    packet[0] = float(local.shape[0])
    k = min(local.shape[0], bound)
    packet[1:1 + k * dim] = torch.from_numpy(local[:k].reshape(-1))
    recv = [torch.zeros_like(packet) for _ in range(world)]
    dist.all_gather(recv, packet)
    counts = [int(p[0].item()) for p in recv]
    if any(c > bound for c in counts):
        return None, counts
    merged = np.concatenate([recv[r][1:1 + counts[r] * dim].numpy().reshape(counts[r], dim) for r in range(world)], axis=0)
    return merged, counts


def allreduce_sum(dist, array):
    """ncclAllReduce(sum) of a float64 array (per-hill integrals / delta grid of the sharded dense path)."""
    import torch

    t = torch.from_numpy(np.ascontiguousarray(array, dtype=np.float64).copy())
    dist.all_reduce(t)
    return t.numpy()
