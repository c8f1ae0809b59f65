"""Deterministic synthetic inputs for the EDM bias hot path.

SplitMix64 with stated seeds, so C++ and Python produce identical doubles via
``(z >> 11) * 2**-53``.  Pure numpy; no oracle, no GPU.
"""
import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed, n, offset=0):
    """n successive SplitMix64 outputs (uint64) of the stream started at ``seed``,
    skipping the first ``offset`` outputs."""
    with np.errstate(over="ignore"):
        k = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = np.uint64(seed) + k * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def uniform(seed, n, offset=0):
    """float64 uniforms in [0, 1)."""
    return (splitmix64(seed, n, offset) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


# ---- C1D: the pairwise-distance CV of fix edm_pair (BASELINE configs[1], [2]) ----
C1D = dict(
    dim=1, lo=[0.0], hi=[2.8], spacing=[0.00025], sigma=[0.025], periodic=[0],
    boundary_lo=[0.0], boundary_hi=[2.8], boundary_periodic=[0],
)
W1_PAIRS = 1 << 20          # 1 048 576 pair distances: the "1M-pair 1D CV" of the metric
W2_PAIRS = 38_800_000       # 1M atoms x 38.8 half-list pairs


def pair_distances(n, seed, offset=0):
    """Uniform-in-shell distances on [0.85, 2.8): an LJ-melt-like g(r) ~ 1."""
    u = uniform(seed, n, offset)
    return np.cbrt(u * (2.8 ** 3 - 0.85 ** 3) + 0.85 ** 3)


# ---- C2D / C3D: coordinate CVs of fix edm (BASELINE configs[3], [4]) ----
C2D = dict(
    dim=2, lo=[0.0, 0.0], hi=[64.0, 64.0], spacing=[1 / 32.0, 1 / 32.0], sigma=[0.125, 0.125],
    periodic=[1, 1], boundary_lo=[0.0, 0.0], boundary_hi=[64.0, 64.0], boundary_periodic=[1, 1],
)
C3D = dict(
    dim=3, lo=[0.0] * 3, hi=[64.0] * 3, spacing=[0.125] * 3, sigma=[0.25] * 3,
    periodic=[1, 1, 1], boundary_lo=[0.0] * 3, boundary_hi=[64.0] * 3, boundary_periodic=[1, 1, 1],
)


def atom_positions(n, seed, box=64.0):
    """[n, 3] float64 positions uniform in [0, box)^3 (LAMMPS atom->x layout)."""
    return (uniform(seed, 3 * n) * box).reshape(n, 3)


def shard_bounds(n, world_size, rank):
    """Contiguous [begin, end) slice of n samples owned by ``rank`` (sizes differ by <= 1)."""
    base, rem = divmod(n, world_size)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)
