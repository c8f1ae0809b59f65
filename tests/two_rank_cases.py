"""Scenarios shared by the ranks (two_rank_worker.py) and the single-process twin (test_gpu_two_ranks.py)."""
import numpy as np

import edm_amd.workloads as W

C1D = "tempering 0\ndimension 1\nbox_low 0\nbox_high 2.8\nbias_spacing 0.001\nbias_sigma 0.05\n"
C2D = "tempering 0\ndimension 2\nbox_low 0 0\nbox_high 8 8\nbias_spacing 0.05 0.05\nbias_sigma 0.2 0.2\n"

CASES = {
    # stochastic steps through the packed exchange (fixed-size packets, one all-gather, device-side unpack, deferred
    # count); step 2 accepts EVERY sample: the packets overflow, the count is poisoned, all ranks fall back together
    # (900 samples per rank: the all-accepted step defers ~1780 hills, inside the 2048-slot overflow buffer, and stays
    #  below the 4096 hills at which the fallback would take the sharded dense application instead)
    "stochastic_1d": dict(cfg=C1D + "hill_prefactor 0.5\nhill_density 60\nbias_per_step 0.3\n", dim=1, n=900, steps=5,
                          lo=[0.0], hi=[2.8], per=[0], skin=[0.3], mode="add_hills", all_accept_step=3, est=1800),
    # the fix edm_pair step: forces ride in the packing launch
    "pair_step_1d": dict(cfg=C1D + "hill_prefactor 0.5\nhill_density 60\nbias_per_step 0.15\n", dim=1, n=3000, steps=4,
                         lo=[0.0], hi=[2.8], per=[0], skin=[0.3], mode="pair_step", all_accept_step=-1, est=6000),
    # all-samples mode: the synchronous record exchange, then the sharded dense application (integral slices
    # all-gathered, delta grids all-reduced); rank 1 owns one hill fewer (unequal slices), the limiter binds
    "dense_1d": dict(cfg=C1D + "hill_prefactor 0.5\nbias_per_step 0.98\n", dim=1, n=4100, steps=3,
                     lo=[0.0], hi=[2.8], per=[0], skin=[0.3], mode="add_hills", all_accept_step=-1, est=4100, uneven=True),
    # a 2-D periodic grid: stochastic steps, the replay scheme
    "stochastic_2d": dict(cfg=C2D + "hill_prefactor 0.5\nhill_density 80\nbias_per_step 0.2\n", dim=2, n=20000, steps=4,
                          lo=[0.0, 0.0], hi=[8.0, 8.0], per=[1, 1], skin=[0.0, 0.0], mode="add_hills", all_accept_step=-1,
                          est=40000),
}


def inputs(case, step, rank):
    """samples and uniforms of one rank for one step (host arrays)"""
    n = case["n"] - (1 if case.get("uneven") and rank == 1 else 0)
    dim = case["dim"]
    if dim == 1:
        x = W.pair_distances(n, 9000 + 100 * step + rank).reshape(-1, 1)
    else:
        x = W.uniform(9000 + 100 * step + rank, n * dim).reshape(n, dim) * np.array(case["hi"])
    u = W.uniform(9500 + 100 * step + rank, n)
    if step == case["all_accept_step"]:
        u = np.zeros(n)
    return np.ascontiguousarray(x), u


def drive(H, b, case, rank, nranks, twin=False):
    """runs the scenario on bias handle b; twin = the single-process run fed the rank-major concatenation"""
    b.setup(1.0, 1.0)
    b.subdivide(case["lo"], case["hi"], case["lo"], case["hi"], case["per"], case["skin"])
    dim = case["dim"]
    energies, forces, state = [], [], []
    for step in range(case["steps"]):
        if twin:
            parts = [inputs(case, step, r) for r in range(nranks)]
            x = np.concatenate([p[0] for p in parts])
            u = np.concatenate([p[1] for p in parts])
        else:
            x, u = inputs(case, step, rank)
        n = len(x)
        d_x = H.DeviceArray.from_host(x)
        d_u = H.DeviceArray.from_host(u)
        has_density = "hill_density" in case["cfg"]
        if case["mode"] == "pair_step":
            d_f = H.DeviceArray.zeros((n,))
            e = b.pair_step_device(d_x, d_f, n, d_x, d_u, n, est=case["est"])
            energies.append(e)
            forces.append(d_f.to_host())
        else:
            b.add_hills_device(d_x, n, dim, d_u if has_density else None, -1, est=case["est"])
        state.append([b.get(k) for k in ("cum_bias", "overflow_left", "overflow_right", "b_skip_hill_add", "hills_added", "steps")])
    v, dv = b.gauss.download()
    out = dict(values=v, derivs=dv, hist=b.hist.values, state=np.array(state), bound_redos=b.get("bound_redos"))
    if energies:
        out["energies"] = np.array(energies)
        out["forces"] = np.concatenate(forces)
    return out
