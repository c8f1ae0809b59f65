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
