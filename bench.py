#!/usr/bin/env python3
"""bench.py -- the EDM per-timestep bias hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input on every GPU:
  (1) bias-force evaluation for W1 = 1,048,576 pair distances of a 1-D r-CV
      (fix edm_pair, BASELINE.json configs[1]): EDMBias::update_force batched, and
  (2) one hill step: EDMBias::add_hills semantics, hill_density 250, 2 add_hill calls
      per pair worth of estimate, bias limiting active (bias_per_step = hill_prefactor),
  in the REFERENCE FIX'S ORDER (lammps/fix_edm_pair.cpp:177-238: pair k's force is read behind the hills of the
  add_hill calls before it -- edm_hip_bias_pair_step_ordered, the default of the rewritten fix edm_pair) and with the
  reference's per-hill HILLS log ON (the library's default).  The batched order (every force on the bias as it
  stands after pre_add_hill: keyword batch_order) and the log-off step are timed beside it.
Inputs are resident in HBM before the timed region.  N > 1 is weak scaling: every rank
(one process per GPU) owns its own W1-sized pair set of the same system, hill_density is
a per-system quantity (divided by the rank count exactly like EDMBias::subdivide does
under MPI) and ranks exchange over RCCL.  value = pairs evaluated by all ranks / time.
With --gpus N > 1 and no WORLD_SIZE in the environment the script starts the N ranks itself
(python -m torch.distributed.run, one process per GPU) before anything touches a GPU.

Rank 0 prints ONE JSON line (see the driver contract) with these extra objects:
  roofline      -- the bias-force kernel of the W1 step: algorithmic 16 B/eval over its HIP-event duration
                   measured on the kernel's own stream inside the timed region; roofline.other holds the same figure
                   for W2 (38.8 M pairs, the HBM-bound capture), for K2 on the 2048^2 and 512^3 coordinate-CV grids,
                   and the metric's second quantity, hill adds/s (1,048,576 hills per step split over the GPUs)
  coordinate_cv -- BASELINE configs[3] / [4] (2048^2 and 512^3 coordinate-CV grids, 262 144 atoms): lookup kernel
                   time, algorithmic bytes per atom, roofline fraction, full fix-edm step (per-step min / median / max,
                   polled / fallback batches, replica rebuilds), the step from HOST arrays (PCIe-inclusive)
  pcie_inclusive -- the same W1 step with pair arrays starting and ending in HOST memory
  cpu_baseline  -- the real reference build (or the CPU oracle) timed on a bounded sample on the host cores
"""
import ctypes as C
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
BYTES_PER_EVAL = 16    # 8 B distance in + 8 B force out (SURVEY.md 8d, K1 r-array form)

CFG = ("tempering 0\nhill_prefactor 0.5\nhill_density 250\ndimension 1\nbox_low 0\nbox_high 2.8\n"
       "bias_spacing 0.00025\nbias_sigma 0.025\n")


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of a kernel from the newest committed rocprofv3 --pmc capture of this
    same command (profiles/rNN_summary.json, written by profiles/summarize.py; FETCH_SIZE and
    WRITE_SIZE collected in separate passes, read side doubled per the gfx950 note of the guide).
    None when no capture is committed: bench.py itself never runs the profiler."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")))
    if not files:
        return None
    try:
        pmc = json.load(open(files[-1])).get("pmc", {})
    except Exception:  # noqa: BLE001
        return None
    for key, e in pmc.items():
        if key.startswith(kernel_prefix) and "hbm_bytes_per_launch" in e:
            return e["hbm_bytes_per_launch"]
    return None


def rocprof_avg_us(kernel_prefix):
    """Average duration of a kernel in the newest committed rocprofv3 --kernel-trace capture of this
    command (profiles/rNN_summary.json) -- printed beside the live figure so the two can be compared."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")))
    if not files:
        return None
    try:
        kernels = json.load(open(files[-1])).get("kernels", [])
    except Exception:  # noqa: BLE001
        return None
    for k in kernels:
        if k["kernel"].startswith(kernel_prefix):
            return k.get("avg_us_timed_region", k["avg_us"])
    return None


def copy_probe(H, src, dst, nbytes, reps=50):
    """Context for the roofline fractions: the runtime's own device-to-device copy (hipMemcpyAsync, a blit kernel)
    moving the SAME traffic as one launch of the lookup kernel (nbytes read + nbytes written), timed with HIP
    events around `reps` back-to-back copies.  Returns microseconds per copy, or None."""
    try:
        hip = C.CDLL("libamdhip64.so")
        ev = [C.c_void_p(), C.c_void_p()]
        for e in ev:
            if hip.hipEventCreate(C.byref(e)) != 0:
                return None
        H.synchronize()
        for _ in range(3):
            hip.hipMemcpyAsync(C.c_void_p(dst), C.c_void_p(src), C.c_size_t(nbytes), 3, None)
        hip.hipEventRecord(ev[0], None)
        for _ in range(reps):
            hip.hipMemcpyAsync(C.c_void_p(dst), C.c_void_p(src), C.c_size_t(nbytes), 3, None)
        hip.hipEventRecord(ev[1], None)
        hip.hipEventSynchronize(ev[1])
        ms = C.c_float(0)
        rc = hip.hipEventElapsedTime(C.byref(ms), ev[0], ev[1])
        for e in ev:
            hip.hipEventDestroy(e)
        return ms.value / reps * 1e3 if rc == 0 else None
    except Exception:  # noqa: BLE001
        return None


def make_bias(mod, tmpdir, tag, rank=0):
    cfg = os.path.join(tmpdir, "bench_%s_%d.edm" % (tag, rank))
    with open(cfg, "w") as fh:
        fh.write(CFG + "hills_filename %s/HILLS_%s\nhistogram_filename %s/HIST_%s_%d\n" % (tmpdir, tag, tmpdir, tag, rank))
    return cfg


def cpu_baseline(tmpdir, seconds=12.0):
    """Times the CPU checker at the reference's per-call granularity on one host core."""
    from oracle import binding as B
    import edm_amd.workloads as W

    kind, lib = "port", None
    try:
        lib = B.load("ref")
        kind = "reference"
    except Exception:  # noqa: BLE001
        lib = B.load("oracle")
    cfg = make_bias(None, tmpdir, "cpu")
    b = B.Bias(lib, cfg)
    b.setup(1.0, 1.0)
    b.subdivide([0], [2.8], [0], [2.8], [0], [0.3])
    g = b.gauss
    for x in W.pair_distances(512, 2):
        g.add_value([x], 1e-3)
    n = 1 << 18
    r = W.pair_distances(n, 1).reshape(-1, 1).copy()
    f = np.zeros_like(r)
    t0 = time.perf_counter()
    evals = 0
    while time.perf_counter() - t0 < seconds * 0.6:
        b.update_forces(r, f)
        evals += n
    t_eval = time.perf_counter() - t0
    # hill adds: GaussGrid::add_value on the C1D stencil (1131 nodes, McGDP boundary), batches of 2048 hills per C call
    hills = W.pair_distances(1 << 20, 5).reshape(-1, 1).copy()
    t0 = time.perf_counter()
    done = 0
    while done + 2048 <= len(hills) and time.perf_counter() - t0 < seconds * 0.4:
        g.add_values(hills[done:done + 2048], 1e-6)
        done += 2048
    t_hill = time.perf_counter() - t0
    return dict(value=evals / t_eval / 1e6, unit="million bias-force evals/s", cores=1, kind=kind,
                hill_adds_per_s=done / t_hill, host_cores_visible=os.cpu_count(),
                sample="%d update_forces passes over %d C1D pair distances (%.1f s) + %d add_value hills in C-side batches of "
                       "2048 (%.1f s), 1 thread" % (evals // n, n, t_eval, done, t_hill))


def cpu_baseline_all_cores(tmpdir, seconds=8.0):
    """The reference's own scaling model on the host: one independent replica per core (MPI ranks each own a shard
    of the samples).  Children are separate interpreters (this process has initialised the GPU) running the
    single-core baseline concurrently; the rates add up."""
    import subprocess

    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(avail, 16))   # (a one-GPU slice of the host: at most 16 replicas; `nproc` is reported beside it)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(seconds),
                               "--cpu-worker-dir", os.path.join(tmpdir, "w%d" % i)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True) for i in range(cores)]
    ev = hl = 0.0
    ok = 0
    for p in procs:
        try:
            out, _ = p.communicate(timeout=seconds * 6 + 60)
            d = json.loads(out.strip().splitlines()[-1])
            ev += d["value"]
            hl += d["hill_adds_per_s"]
            ok += 1
        except Exception:  # noqa: BLE001
            p.kill()
    if ok == 0:
        return None
    return dict(value=ev, unit="million bias-force evals/s", cores=ok, nproc=avail, hill_adds_per_s=hl,
                sample="%d concurrent single-core replicas of the baseline above (nproc = %d), %.0f s each" % (ok, avail, seconds))


def all_samples_measure(rank, world, dist, H, W, tmpdir, steps, warmup):
    """Strong scaling of the all-samples hill step: 1,048,576 hills per step in total, each rank
    owns 1/N of the samples, computes their integrals and gathers them into its delta grid; integrals and
    delta grids are all-reduced (RCCL) and every rank adds the same total to its replica.  Returns
    (hills per step in total, seconds per step as the max over ranks, cum_bias)."""
    total = W.W1_PAIRS
    n = total // world
    cfg = os.path.join(tmpdir, "bench_all_%d.edm" % rank)
    with open(cfg, "w") as fh:
        fh.write(CFG.replace("hill_density 250\n", "").replace("hill_prefactor 0.5", "hill_prefactor 1e-3")
                 + "bias_per_step 1e9\nhills_filename %s/HILLS_all\nhistogram_filename %s/HIST_all_%d\n" % (tmpdir, tmpdir, rank))
    b = H.Bias(cfg)
    if dist is not None:
        ident = [H.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ident, src=0)
        b.comm_init(ident[0], world, rank)
    b.setup(1.0, 1.0)
    b.subdivide([0], [2.8], [0], [2.8], [0], [0.3])
    # (the one place the per-hill HILLS log is switched off: a million hills per step are 80 MB of text per step --
    #  the reference writes them, at ~1 us of printf per line; every other section runs with the log on, the default)
    b.set_hill_log(False)
    r = W.pair_distances(total, 1)[rank * n:(rank + 1) * n]
    d_r = H.DeviceArray.from_host(np.ascontiguousarray(r))

    def barrier():
        H.synchronize()
        if dist is not None:
            dist.barrier()
            H.synchronize()

    for _ in range(warmup):
        b.add_hills_device(d_r, n, 1, None, -1, n)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        b.add_hills_device(d_r, n, 1, None, -1, n)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch

        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    cum = b.get("cum_bias")
    del b
    return n * world, elapsed / steps, cum


def all_samples_line(args, rank, world, dist, H, W, tmpdir):
    hills, sec, cum = all_samples_measure(rank, world, dist, H, W, tmpdir, args.steps, args.warmup)
    if rank == 0:
        print(json.dumps({
            "metric": "hill adds/sec (1M-pair 1D CV, all-samples mode: every pair deposits a hill each step)",
            "value": hills / sec, "unit": "hill adds/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": sec * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "W1 all-samples: %d pair distances in total, C1D grid (11201 nodes, stencil 1131, McGDP "
                                   "boundary), hill_density unset, bias_per_step not binding" % hills,
                       "hills_per_gpu": hills // world,
                       "parallelism": "replicated grid, hills sharded, integrals + delta grid all-reduced, dp%d" % world},
            "cum_bias": cum}))
    if dist is not None:
        dist.destroy_process_group()


def headline_dict(args, world, npairs, elapsed, k_ms, k_launches, timed_every, tail_us=()):
    """The contract keys of the JSON line, complete once the timed region has ended."""
    ms_per_step = elapsed / args.steps * 1e3
    total_pairs = npairs * world
    # the stamped launches: those of the timed region (their mean) and the stamped tail right behind it (see main());
    # the median over all of them is the launch's duration -- one sample of a 12 us launch is what the driver's
    # --steps 20 leaves in the timed region, and a single stamp read 20 us once where every other read 11.5
    timed_us = k_ms / max(k_launches, 1) * 1e3
    samples = ([timed_us] if k_launches else []) + list(tail_us)
    samples.sort()
    k_s = (samples[len(samples) // 2] if samples else timed_us) * 1e-6
    # 16 B per evaluation (8 B distance in + 8 B force out) x the pairs one launch processes.  The
    # reference-order launch also reads one 4-B sample index per pair (where the pair's add_hill calls begin) -- real
    # traffic of the same launch, reported beside the 8(d) figure, not inside it
    alg_bytes = BYTES_PER_EVAL * npairs
    achieved = alg_bytes / k_s / 1e9
    with_u = (alg_bytes + 4 * npairs) / k_s / 1e9
    return {
        "metric": "million bias-force evals/sec (1M-pair 1D CV, force eval + hill step per step, reference fix's order, HILLS log on)",
        "value": total_pairs / (elapsed / args.steps) / 1e6,
        "unit": "million evals/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "W1: fix edm_pair 1-D r-CV, %d pair distances per GPU (32k-atom LJ melt ~ 1M half-list pairs), "
                        "grid 0..2.8 spacing 0.00025 (11201 nodes) sigma 0.025 McGDP boundary, hill_density 250, "
                        "bias_per_step = hill_prefactor, bias pre-populated with 4096 hills" % npairs,
            "pairs_per_gpu": npairs,
            "hill_step_every": 1,
            "order": "reference (lammps/fix_edm_pair.cpp:177-238: pair k's force behind the hills of pairs 0..k-1; the fix's default)",
            "hills_log": "on (edm_bias.cpp:586-599, the library's default)",
            "parallelism": "replicated bias grid, samples sharded, dp%d" % world,
        },
        "roofline": {
            "kernel": "k_pair_forces_ordered (K1 in the reference fix's order: each pair's two corner records as they stood behind "
                      "the hills of the add_hill calls before it)",
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": pmc_traffic("edm::k_pair_forces_ordered"),
            "kernel_us": k_s * 1e6,
            "kernel_us_timed_region": timed_us,
            "kernel_us_samples": [round(v, 2) for v in samples],
            "kernel_us_note": "median over the stamped launches: the timed region's (mean of `launches`) and a tail of stamped "
                              "steps of the same loop right behind it",
            "kernel_us_rocprof": rocprof_avg_us("edm::k_pair_forces_ordered"),
            "launches": k_launches,
            "timed_every": timed_every,
            "bytes_per_launch": alg_bytes,
            "bytes_per_launch_note": "16 B per pair (SURVEY 8d: distance in + force out)",
            "achieved_incl_sample_indices": with_u,
            "frac_incl_sample_indices": with_u / HBM_PEAK_GBS,
            "bytes_per_launch_incl_sample_indices": alg_bytes + 4 * npairs,
        },
    }


class ExtrasGuard:
    """Prints the line exactly once: the full one when the informational extras are done, or -- from a timer thread,
    should they not finish within limit_s -- the headline alone, naming the extra that was running, and ends the
    process with status 3 (a stalled extra is a failure of the run, not a clean finish).  The limit is a backstop
    for a hang nobody has seen; the extras of a default run take well under a minute."""

    EXIT_STALLED = 3

    def __init__(self, rank, headline, limit_s):
        import threading

        self.rank, self.headline, self.limit_s = rank, headline, limit_s
        self.lock = threading.Lock()
        self.done = False
        self.current = "none"
        # every rank gives up at the same moment: no rank is left inside a collective its peers have abandoned for
        # longer than the peers' own exit takes
        self.timer = threading.Timer(limit_s, self._expired)
        self.timer.daemon = True
        self.timer.start()

    def stage(self, name):
        self.current = name

    def _expired(self):
        with self.lock:
            if self.done:
                return
            self.done = True
            if self.rank == 0:
                out = dict(self.headline)
                out["extras_skipped"] = ("the informational extra '%s' did not finish within %.0f s; the keys above were "
                                         "measured before it; exit status %d" % (self.current, self.limit_s, self.EXIT_STALLED))
                print(json.dumps(out), flush=True)
        os._exit(self.EXIT_STALLED)

    def finish(self, out):
        with self.lock:
            if self.done:
                return
            self.done = True
            self.timer.cancel()
            if out is not None:
                print(json.dumps(out), flush=True)


def self_launch(args):
    """--gpus N > 1 without a launcher: start the N ranks as fresh child processes (one per GPU, RCCL over xGMI) and
    return their exit status.  Nothing in this process has touched a GPU (counting devices does not)."""
    import socket
    import subprocess

    import torch

    ndev = torch.cuda.device_count()
    if ndev < args.gpus:
        print("bench.py: --gpus %d but only %d GPU(s) are visible" % (args.gpus, ndev), file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def coordinate_cv_measure(H, W, tmpdir):
    """BASELINE configs[3] / [4]: the coordinate-CV grids (2048^2, 512^3), 262 144 atoms at random positions resident
    in HBM.  Lookup kernel timed by its own dispatch timestamps; algorithmic bytes per atom from SURVEY 8(d)."""
    nd = {}
    for tag, c, per_atom in (("c2d_2048sq", W.C2D, 156), ("c3d_512cube", W.C3D, 332)):
        gg = H.Gauss.create(c["lo"], c["hi"], c["spacing"], c["periodic"], 1, c["sigma"])
        dim = c["dim"]
        natoms = 262144
        x = W.atom_positions(natoms, 21 if dim == 2 else 31)
        d_x = H.DeviceArray.from_host(x)
        d_ff = H.DeviceArray.zeros((natoms, 3))
        d_uu = H.DeviceArray.from_host(W.uniform(77, natoms))
        hills = H.DeviceArray.from_host(np.ascontiguousarray(x[:250]))
        tot = H.C.c_double(0)
        H.check(H.lib().edm_hip_gauss_add_values(gg.h, 250, hills.ptr, 3, None, 0.01, None, H.C.byref(tot)))
        e = H.C.c_double(0)
        H.check(H.lib().edm_hip_gauss_update_forces(gg.h, natoms, d_x.ptr, 3, d_ff.ptr, 3, None, -1, H.C.byref(e)))
        gg.profile_enable(True)
        gg.profile_read(reset=True)
        for _ in range(20):
            H.check(H.lib().edm_hip_gauss_update_forces(gg.h, natoms, d_x.ptr, 3, d_ff.ptr, 3, None, -1, H.C.byref(e)))
        ms, ln = gg.profile_read(reset=True)
        # the same atoms in LAMMPS' default memory order: spatially sorted into bins of half the neighbour
        # cutoff (atom_modify sort, binsize 1.4)
        nb = int(np.ceil(64.0 / 1.4))
        bins = np.floor(x / 1.4).astype(np.int64)
        order = np.argsort(bins[:, 0] + nb * (bins[:, 1] + nb * bins[:, 2]), kind="stable")
        d_xs = H.DeviceArray.from_host(np.ascontiguousarray(x[order]))
        H.check(H.lib().edm_hip_gauss_update_forces(gg.h, natoms, d_xs.ptr, 3, d_ff.ptr, 3, None, -1, H.C.byref(e)))
        gg.profile_read(reset=True)
        for _ in range(20):
            H.check(H.lib().edm_hip_gauss_update_forces(gg.h, natoms, d_xs.ptr, 3, d_ff.ptr, 3, None, -1, H.C.byref(e)))
        ms_s, ln_s = gg.profile_read(reset=True)
        gg.profile_enable(False)
        H.synchronize()
        t3 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            H.check(H.lib().edm_hip_gauss_add_values(gg.h, 250, hills.ptr, 3, None, 0.01, None, H.C.byref(tot)))
        H.synchronize()
        t_h = (time.perf_counter() - t3) / reps
        replica = gg.lookup_replica_info()
        gbs = per_atom * natoms / (ms / ln * 1e-3) / 1e9
        # one hill-depositing fix edm step on this grid (W3 / W4 of SURVEY 8d): update_forces over all atoms +
        # add_hills (hill_density 250; W4: bias_per_step = 0.4 x the expected per-step sum, so the limiter and
        # the overflow buffer work every step) through edm_hip_bias_step, atoms resident in HBM
        cfgp = os.path.join(tmpdir, "bench_%s.edm" % tag)
        with open(cfgp, "w") as fh:
            fh.write("tempering 0\nhill_prefactor %g\nhill_density 250\n%sdimension %d\nbox_low %s\nbox_high %s\n"
                     "bias_spacing %s\nbias_sigma %s\nhills_filename %s/HILLS_%s\nhistogram_filename %s/HIST_%s\n" % (
                         0.02 if dim == 3 else 0.5, "bias_per_step 0.008\n" if dim == 3 else "", dim,
                         " ".join("0" for _ in range(dim)), " ".join("64" for _ in range(dim)),
                         " ".join("%.10g" % v for v in c["spacing"]), " ".join("%.10g" % v for v in c["sigma"]),
                         tmpdir, tag, tmpdir, tag))
        del gg
        bb = H.Bias(cfgp)
        bb.setup(1.0, 1.0)
        bb.subdivide([0.0] * dim, [64.0] * dim, [0.0] * dim, [64.0] * dim, [1] * dim, [0.0] * dim)
        d_fs = H.DeviceArray.zeros((natoms, 3))
        for _ in range(3):
            bb.step_device(d_x, 3, d_fs, 3, natoms, d_uu, -1, natoms)
        H.synchronize()
        pf0, pb0 = bb.get("poll_fallbacks"), bb.get("polled_batches")
        rb0 = bb.gauss.lookup_replica_info()[2]
        per_step = []
        t4 = time.perf_counter()
        for _ in range(40):
            t5 = time.perf_counter()
            bb.step_device(d_x, 3, d_fs, 3, natoms, d_uu, -1, natoms)
            per_step.append(time.perf_counter() - t5)
        H.synchronize()
        t_step = (time.perf_counter() - t4) / 40
        per_step.sort()
        step_stats = dict(min_ms=per_step[0] * 1e3, median_ms=per_step[len(per_step) // 2] * 1e3, max_ms=per_step[-1] * 1e3,
                          poll_fallbacks=bb.get("poll_fallbacks") - pf0, polled_batches=bb.get("polled_batches") - pb0,
                          replica_rebuilds=bb.gauss.lookup_replica_info()[2] - rb0,
                          note="host clock around each of the 40 calls (a call returns when its results are on the host; the "
                               "grid update may still run); a poll fallback = a batch whose completion word did not arrive "
                               "within 2 ms and was waited for on the stream")
        hills_step = bb.get("hills_added") / 43.0
        # the same step as the fix pays for it: atom->x / atom->f in HOST memory (edm_hip_bias_step_host: positions up,
        # the bias-force delta down and added on the host; the force array is never uploaded)
        xh = np.ascontiguousarray(x)
        fh = np.zeros((natoms, 3))
        uh = W.uniform(77, natoms)
        for _ in range(3):
            bb.step_host(xh, fh, runiform=uh, apply_mask=-1, hill_step=True, est=natoms)
        H.synchronize()
        t6 = time.perf_counter()
        for _ in range(20):
            bb.step_host(xh, fh, runiform=uh, apply_mask=-1, hill_step=True, est=natoms)
        H.synchronize()
        t_host = (time.perf_counter() - t6) / 20
        # ... against plain copies of the same bytes (24 B per atom up, 8 * dim down)
        hip = C.CDLL("libamdhip64.so")
        d_tmp = H.DeviceArray((natoms, 3))
        t7 = time.perf_counter()
        rc_copy = 0
        for _ in range(20):
            # (hipMemcpyDefault: the position block is page-locked in place by now, and an explicit host-to-device kind
            #  on registered memory is refused as an invalid value)
            rc_copy |= hip.hipMemcpy(C.c_void_p(d_tmp.ptr), C.c_void_p(xh.ctypes.data), C.c_size_t(xh.nbytes), 4)
            rc_copy |= hip.hipMemcpy(C.c_void_p(fh.ctypes.data), C.c_void_p(d_tmp.ptr), C.c_size_t(8 * dim * natoms), 4)
        t_copy = (time.perf_counter() - t7) / 20
        if rc_copy:
            raise RuntimeError("plain hipMemcpy probe failed: %d" % rc_copy)
        # ... and against 24 B per atom in EACH direction (what a fix that ships atom->x up and atom->f down moves)
        t8 = time.perf_counter()
        for _ in range(20):
            rc_copy |= hip.hipMemcpy(C.c_void_p(d_tmp.ptr), C.c_void_p(xh.ctypes.data), C.c_size_t(xh.nbytes), 4)
            rc_copy |= hip.hipMemcpy(C.c_void_p(fh.ctypes.data), C.c_void_p(d_tmp.ptr), C.c_size_t(24 * natoms), 4)
        t_copy48 = (time.perf_counter() - t8) / 20
        if rc_copy:
            raise RuntimeError("plain hipMemcpy probe failed: %d" % rc_copy)
        pcie = dict(ms_per_step=t_host * 1e3, bytes_up_per_atom=24 + 8, bytes_down_per_atom=8 * dim,
                    plain_copies_of_the_same_bytes_ms=t_copy * 1e3, ratio_to_plain_copies=t_host / t_copy,
                    plain_copies_of_2x24B_per_atom_ms=t_copy48 * 1e3, ratio_to_copies_of_2x24B_per_atom=t_host / t_copy48,
                    note="edm_hip_bias_step_host on pageable numpy arrays (the position block is page-locked in place by the "
                         "library): positions + uniforms up, bias-force delta down, added to the force array on the host")
        del bb
        kname = "edm::k_lookup_quad<%d" % dim
        nd[tag] = dict(workload="BASELINE configs[%d]: fix edm coordinate CV, %s periodic bias grid, %d atoms at random positions"
                                % (3 if dim == 2 else 4, "2048^2" if dim == 2 else "512^3", natoms),
                       atoms=natoms,
                       roofline=dict(kernel="k_lookup_quad<%d, forces> (K2 on the lookup replica, four lanes per atom)" % dim,
                                     bound="hbm", achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS,
                                     traffic=pmc_traffic(kname), kernel_us=ms / ln * 1e3,
                                     kernel_us_rocprof=rocprof_avg_us(kname), launches=ln,
                                     bytes_per_launch=per_atom * natoms,
                                     bytes_per_launch_note="%d B per atom (SURVEY 8d: position row + mask + force RMW + %d corner records)"
                                                           % (per_atom, 2 ** dim)),
                       million_atom_evals_per_s=natoms / (ms / ln * 1e-3) / 1e6,
                       lookup_replica=dict(in_use=replica[0], bytes=replica[1]),
                       lookup_kernel_us_bin_sorted_atoms=ms_s / ln_s * 1e3,
                       frac_of_hbm_peak_bin_sorted_atoms=per_atom * natoms / (ms_s / ln_s * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       step_ms=t_step * 1e3, step_million_atom_evals_per_s=natoms / t_step / 1e6,
                       step_stats=step_stats, pcie_inclusive_step=pcie,
                       step_hills_added_avg=hills_step,
                       hill_batch_250_ms=t_h * 1e3, hill_adds_per_s=250 / t_h)
    return nd


def pcie_inclusive_measure(H, b, r, u, npairs, est, steps=20):
    """The W1 step as the host-list `fix edm_pair` pays for it: pair distances and sample uniforms in, pair forces out
    cross PCIe every step.  Three ways: synchronous copies around edm_hip_bias_pair_step (round 1's libedm.so),
    edm_hip_bias_pair_step_host on pageable arrays, and on page-locked arrays (what EDMBias::pair_step and the
    fix's pinned vectors do now: copies queued around the kernels, the forces travelling down while the uniforms
    travel up).  Never the headline `value`."""
    d_r = H.DeviceArray((npairs,))
    d_u = H.DeviceArray((npairs,))
    d_f = H.DeviceArray((npairs,))
    f_host = np.empty(npairs)

    def sync_copies():
        H.check(H.lib().edm_hip_memcpy_h2d(d_r.ptr, r.ctypes.data, r.nbytes))
        H.check(H.lib().edm_hip_memcpy_h2d(d_u.ptr, u.ctypes.data, u.nbytes))
        e = b.pair_step_device(d_r, d_f, npairs, d_r, d_u, npairs, est)
        H.check(H.lib().edm_hip_memcpy_d2h(f_host.ctypes.data, d_f.ptr, f_host.nbytes))
        return e

    p_r, p_u, p_f = H.pinned_array(npairs), H.pinned_array(npairs), H.pinned_array(npairs)
    p_r[:] = r
    p_u[:] = u

    def timed(fn):
        for _ in range(3):
            fn()
        H.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        H.synchronize()
        return (time.perf_counter() - t0) / steps

    first = np.arange(npairs, dtype=np.int32)
    dt_sync = timed(sync_copies)
    dt_pageable = timed(lambda: b.pair_step_host(r, f_host, r, u, est))
    dt_pinned_batch = timed(lambda: b.pair_step_host(p_r, p_f, p_r, p_u, est))
    dt_pinned = timed(lambda: b.pair_step_ordered_host(p_r, p_f, first, p_r, p_u, est))
    return dict(ms_per_step=dt_pinned * 1e3, million_evals_per_s=npairs / dt_pinned / 1e6,
                bytes_over_pcie_per_step=(3 * 8 + 4) * npairs,
                effective_GBs=(3 * 8 + 4) * npairs / dt_pinned / 1e9,
                ms_per_step_batch_order=dt_pinned_batch * 1e3,
                ms_per_step_batch_order_pageable_arrays=dt_pageable * 1e3,
                ms_per_step_batch_order_synchronous_copies=dt_sync * 1e3,
                note="8 B distance + 8 B uniform + 4 B sample index in, 8 B force out per pair, per step; ms_per_step = "
                     "edm_hip_bias_pair_step_ordered_host (the fix's default order) on page-locked host arrays; batch order: "
                     "edm_hip_bias_pair_step_host (copies queued around the kernels, forces down while uniforms go up)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--pairs", type=int, default=0, help="pairs per GPU (default W1 = 1,048,576)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-w2", action="store_true",
                    help="skip the 38.8M-pair interpolation capture (W2 = BASELINE configs[2], the HBM-bound case)")
    ap.add_argument("--w2", action="store_true", help=argparse.SUPPRESS)  # kept for older command lines (now the default)
    ap.add_argument("--no-nd", action="store_true",
                    help="skip the coordinate-CV section (BASELINE configs[3]/[4]: 2048^2 and 512^3 grids; part of the default line)")
    ap.add_argument("--nd", action="store_true", help=argparse.SUPPRESS)  # kept for older command lines (now the default)
    ap.add_argument("--cpu-worker", type=float, default=0.0, help=argparse.SUPPRESS)   # (child of cpu_baseline_all_cores)
    ap.add_argument("--cpu-worker-dir", default="", help=argparse.SUPPRESS)
    ap.add_argument("--all-samples", action="store_true",
                    help="alternative line: STRONG scaling of the all-samples hill mode (hill_density unset; every one of "
                         "1,048,576 pair distances deposits a hill each step; the samples are split over the GPUs)")
    args = ap.parse_args()

    if args.cpu_worker > 0:   # CPU only: never touches the GPU
        os.makedirs(args.cpu_worker_dir, exist_ok=True)
        print(json.dumps(cpu_baseline(args.cpu_worker_dir, args.cpu_worker)))
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: be the launcher (fresh child processes, before any GPU call in this one)
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    dist = None
    force_dist = os.environ.get("EDM_BENCH_FORCE_DIST") == "1"  # exercise the N>1 plumbing with one rank
    if world > 1 or force_dist:
        import torch
        import torch.distributed as dist_mod

        torch.cuda.set_device(local_rank)
        dist_mod.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        dist = dist_mod

    import edm_amd.hip as H
    import edm_amd.workloads as W

    H.require_gpu()
    H.check(H.lib().edm_hip_set_device(local_rank))
    tmpdir = tempfile.mkdtemp(prefix="edm_bench_")
    npairs = args.pairs or W.W1_PAIRS

    if args.all_samples:
        return all_samples_line(args, rank, world, dist, H, W, tmpdir)

    b = H.Bias(make_bias(H, tmpdir, "gpu", rank))
    if dist is not None:
        import torch

        ident = [H.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ident, src=0)
        b.comm_init(ident[0], world, rank)
    b.setup(1.0, 1.0)
    b.subdivide([0], [2.8], [0], [2.8], [0], [0.3])
    # (the reference's per-hill HILLS log stays ON, the library's default: written by a thread of its own)
    g = b.gauss
    # bias pre-populated with 4096 hills (seed 2), identical on every rank
    hills0 = np.zeros((4096, 1))
    hills0[:, 0] = W.pair_distances(4096, 2)
    g.add_values(hills0, 1e-3)

    # this rank's pair set, resident in HBM
    r = W.pair_distances(npairs, 1 + 1000 * rank)
    u = W.uniform(3 + 1000 * rank, npairs)
    d_r = H.DeviceArray.from_host(r)
    d_u = H.DeviceArray.from_host(u)
    d_f = H.DeviceArray.zeros((npairs,))
    est = 2 * npairs  # fix_edm_pair makes up to two add_hill calls per pair (fix_edm_pair.cpp:230-237)

    # one add_hill sample per pair here: pair k's first (only) sample is sample k
    d_first = H.DeviceArray.from_host(np.arange(npairs, dtype=np.int32))

    def step():
        # one hill-depositing fix edm_pair step IN THE REFERENCE FIX'S ORDER (lammps/fix_edm_pair.cpp:173-247):
        # pre_add_hill(est); per pair update_force, then its add_hill(r, u); post_add_hill -- a single C-ABI call
        # (edm_hip_bias_pair_step_ordered, what the rewritten fix edm_pair calls by default)
        return b.pair_step_ordered_device(d_r, d_f, d_first, npairs, d_r, d_u, npairs, est)

    def step_batch():
        # the same step with every force evaluated on the bias as it stands after pre_add_hill (keyword batch_order)
        return b.pair_step_device(d_r, d_f, npairs, d_r, d_u, npairs, est)

    def barrier():
        H.synchronize()
        if dist is not None:
            dist.barrier()
            H.synchronize()

    for _ in range(args.warmup):
        step()
    # K1's launch is stamped (dispatch begin / end timestamps) on a few steps of the timed region -- 1 of 20, 10 of 200:
    # a stamped launch costs the stream ~15 us (an unstamped step takes ~31 us; every 4th step stamped, round 1's
    # choice, added 3.5 us to EVERY step of the average), so a sparse sample keeps `ms_per_step` what the step costs.
    # (EDM_BENCH_TIMED_EVERY overrides; the first stamped launch of a process pays for switching the queue's profiling
    #  on -- spent here on two extra warm-up steps, outside the timed region)
    n_stamps = min(10, max(1, args.steps // 20))
    TIMED_EVERY = int(os.environ.get("EDM_BENCH_TIMED_EVERY", "0")) or max(1, args.steps // n_stamps)
    g.profile_enable(1)
    for _ in range(2):
        step()
    g.profile_read(reset=True)
    g.profile_enable(TIMED_EVERY)
    g.profile_read(reset=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        energy = step()
    barrier()
    elapsed = time.perf_counter() - t0
    k_ms, k_launches = g.profile_read(reset=True)
    # stamped tail: the same step, every launch of K1's kernel stamped and read one by one (outside the timed region --
    # a stamped launch costs the stream ~15 us)
    tail_us = []
    g.profile_enable(1)
    for _ in range(12):
        step()
        ms_t, l_t = g.profile_read(reset=True)
        if l_t:
            tail_us.append(ms_t / l_t * 1e3)
    g.profile_enable(False)
    if dist is not None:
        import torch

        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # Everything below is informational (component rates, the second quantity of the metric, the HBM-bound capture,
    # the CPU baseline).  The headline numbers are complete at this point: if an extra ever failed to finish -- the
    # multi-rank ones run collectives -- rank 0 still prints the line, marked, instead of losing the measurement.
    gate_giveups = b.get("ord_gate_giveups")
    headline = headline_dict(args, world, npairs, elapsed, k_ms, k_launches, TIMED_EVERY, tail_us)
    # (the reference-order step's record and force pass run on a second stream behind a gate wave that waits for the hill
    #  batch's limiter; a gate that gave up -- kernels of different streams run one at a time, e.g. under rocprofv3 --pmc --
    #  sends the object back to one stream for good: 0 in an ordinary run)
    headline["second_stream_gate_giveups"] = gate_giveups
    # second quantity of the metric (BASELINE.json: "... + hill-adds/sec", target: strong scaling at 8 GPUs): the
    # all-samples hill mode, 1,048,576 hills per step in total split over the GPUs (a collective: every rank runs it);
    # part of the measured line, not of the guarded extras
    hs_hills, hs_sec, _ = all_samples_measure(rank, world, dist, H, W, tmpdir, steps=5, warmup=2)
    headline["hill_adds_strong_scaling"] = dict(
        value=hs_hills / hs_sec, unit="hill adds/s", hills_per_step_total=hs_hills, n_gpus=world, ms_per_step=hs_sec * 1e3,
        scaling="strong", steps=5, warmup=2,
        note="all-samples mode, hills sharded over the GPUs, per-hill integrals + delta grid exchanged over RCCL; same "
             "quantity as the main line of `bench.py --all-samples`")
    guard = ExtrasGuard(rank, headline, limit_s=float(os.environ.get("EDM_BENCH_EXTRAS_LIMIT", "300")))
    guard.stage("device_rng_step")

    # the same step with the acceptance uniforms drawn on the device (fast mode of the fixes' RNG: no array of
    # uniforms is generated, uploaded or read); informational, not part of `value`
    b.set_device_rng(True, 12345 + rank)
    for _ in range(3):
        b.pair_step_device(d_r, d_f, npairs, d_r, None, npairs, est)
    barrier()
    t_r = time.perf_counter()
    for _ in range(args.steps):
        b.pair_step_device(d_r, d_f, npairs, d_r, None, npairs, est)
    barrier()
    ms_step_device_rng = (time.perf_counter() - t_r) / args.steps * 1e3
    b.set_device_rng(False, 0)

    # the other ways to run the step, each timed like the headline (warm-up, steps between barriers), unstamped:
    # batch order (keyword batch_order), and both orders without the HILLS log
    guard.stage("step_modes")

    def timed_steps(fn, steps):
        for _ in range(3):
            fn()
        barrier()
        t_o = time.perf_counter()
        for _ in range(steps):
            fn()
        barrier()
        return (time.perf_counter() - t_o) / steps * 1e3

    ms_ordered_unstamped = timed_steps(step, args.steps)
    ms_batch = timed_steps(step_batch, args.steps)
    b.set_hill_log(False)
    ms_ordered_nolog = timed_steps(step, args.steps)
    ms_batch_nolog = timed_steps(step_batch, args.steps)
    b.set_hill_log(True)
    step_modes = dict(
        ms_per_step_reference_order=ms_ordered_unstamped, ms_per_step_batch_order=ms_batch,
        ms_per_step_reference_order_hills_log_off=ms_ordered_nolog, ms_per_step_batch_order_hills_log_off=ms_batch_nolog,
        note="reference order = edm_hip_bias_pair_step_ordered (the headline's step, here without dispatch stamps); batch order = "
             "edm_hip_bias_pair_step (every force on the bias as it stands after pre_add_hill: fix keyword batch_order; its "
             "hill-step forces differ from the reference's, INTEGRATION.md); HILLS log on unless stated")

    # BASELINE configs[1] end to end from POSITIONS: 32k atoms at the LJ-melt density, half neighbour list within
    # r_c + skin = 2.8 resident on the GPU (fix edm_pair ... gpu_list), every step deposits hills; informational
    lj = None
    guard.stage("lj_melt_32k_from_positions")
    if rank == 0:
        try:
            from scipy.spatial import cKDTree

            na = 32000
            box = (na / 0.8442) ** (1.0 / 3.0)
            xa = W.uniform(5, 3 * na).reshape(na, 3) * box
            pr = cKDTree(xa).query_pairs(2.8, output_type="ndarray")
            pr = pr[np.lexsort((pr[:, 1], pr[:, 0]))].astype(np.int32)
            bl = H.Bias(make_bias(H, tmpdir, "lj", rank))
            bl.setup(1.0, 1.0)
            bl.subdivide([0], [2.8], [0], [2.8], [0], [0.3])
            bl.set_device_rng(True, 777)
            bl.set("reference_order", 1)   # (the fix's default order; gpu_list keeps the neighbour list on the GPU)
            bl.pair_list_upload(pr[:, 0], pr[:, 1], np.ones(na, dtype=np.int32))
            d_xa = H.DeviceArray.from_host(xa)
            d_fa = H.DeviceArray.zeros((na, 3))
            calls = 2 * len(pr)
            for _ in range(3):
                _, calls = bl.pair_list_step_device(na, 1, 1, d_xa, d_fa, True, calls)
            H.synchronize()
            t_l = time.perf_counter()
            for _ in range(50):
                _, calls = bl.pair_list_step_device(na, 1, 1, d_xa, d_fa, True, calls)
            H.synchronize()
            dt_l = (time.perf_counter() - t_l) / 50
            lj = dict(atoms=na, list_entries=int(len(pr)), ms_per_step=dt_l * 1e3,
                      million_pair_evals_per_s=len(pr) / dt_l / 1e6,
                      note="positions resident in HBM; distances, lookups, per-atom force sums and the hill step on the GPU")
            del bl
        except Exception as exc:  # noqa: BLE001  (scipy missing: skip the extra)
            lj = dict(skipped=repr(exc))

    guard.stage("component_rates")
    copy_us = copy_probe(H, d_r.ptr, d_f.ptr, 8 * npairs) if rank == 0 else None
    # component rates (not part of `value`): force evaluation alone, all-samples hill adds
    reps = 20
    H.synchronize()
    t1 = time.perf_counter()
    for _ in range(reps):
        b.pair_forces_device(d_r, d_f, npairs)
    H.synchronize()
    t_eval = (time.perf_counter() - t1) / reps
    extra = {}
    if rank == 0:
        nh = 1 << 18
        hx = H.DeviceArray.from_host(W.pair_distances(nh, 9))
        tot = H.C.c_double(0)
        H.check(H.lib().edm_hip_gauss_add_values(g.h, nh, hx.ptr, 1, None, 1e-9, None, H.C.byref(tot)))
        H.synchronize()
        t2 = time.perf_counter()
        H.check(H.lib().edm_hip_gauss_add_values(g.h, nh, hx.ptr, 1, None, 1e-9, None, H.C.byref(tot)))
        H.synchronize()
        extra["hill_adds_per_s_all_samples"] = nh / (time.perf_counter() - t2)
        extra["hill_adds_sample"] = "%d add_value hills in one batch, C1D stencil 1131 nodes, fused gather + integrals pass" % nh
    pcie = None
    if rank == 0 and dist is None:   # (with a communicator pair_step is a collective: single-GPU runs only)
        guard.stage("pcie_inclusive")
        pcie = pcie_inclusive_measure(H, b, r, u, npairs, est)
    roof_w2 = None
    guard.stage("w2_interpolation_capture")
    if not args.no_w2 and rank == 0:
        n2 = W.W2_PAIRS
        d_r2 = H.DeviceArray.from_host(W.pair_distances(n2, 11))
        d_f2 = H.DeviceArray((n2,))
        g.pair_forces_device(d_r2, d_f2, n2)
        g.profile_enable(True)
        g.profile_read(reset=True)
        for _ in range(10):
            g.pair_forces_device(d_r2, d_f2, n2)
        ms2, l2 = g.profile_read(reset=True)
        g.profile_enable(False)
        a2 = BYTES_PER_EVAL * n2 / (ms2 / l2 * 1e-3) / 1e9
        cp2 = copy_probe(H, d_r2.ptr, d_f2.ptr, 8 * n2, reps=10)
        roof_w2 = dict(workload="W2: %d pair distances" % n2, bound="hbm", achieved=a2, peak=HBM_PEAK_GBS, unit="GB/s",
                       device_copy_same_traffic_us=cp2,
                       frac=a2 / HBM_PEAK_GBS, kernel_ms=ms2 / l2, kernel="k_pair_forces_fast<true> (LDS-staged window)",
                       bytes_per_launch=BYTES_PER_EVAL * n2, traffic=pmc_traffic("edm::k_pair_forces_fast<true"),
                       kernel_ms_rocprof=(rocprof_avg_us("edm::k_pair_forces_fast<true") or 0) / 1e3 or None)
        # the same 38.8 M pairs through the REFERENCE-ORDER force pass (k_pair_forces_ordered, the headline's dominant
        # kernel, out of its latency-bound regime): a hill step whose hills come from the W1 samples, pair k's first
        # add_hill call spread evenly over them
        if dist is None:
            d_first2 = H.DeviceArray.from_host((np.arange(n2, dtype=np.int64) * npairs // n2).astype(np.int32))
            for _ in range(2):
                b.pair_step_ordered_device(d_r2, d_f2, d_first2, n2, d_r, d_u, npairs, est)
            g.profile_enable(True)
            g.profile_read(reset=True)
            for _ in range(6):
                b.pair_step_ordered_device(d_r2, d_f2, d_first2, n2, d_r, d_u, npairs, est)
            ms3, l3 = g.profile_read(reset=True)
            g.profile_enable(False)
            if l3:
                a3 = BYTES_PER_EVAL * n2 / (ms3 / l3 * 1e-3) / 1e9
                roof_w2["reference_order"] = dict(
                    kernel="k_pair_forces_ordered_win (the LDS-window form long arrays take)", kernel_ms=ms3 / l3, achieved=a3,
                    frac=a3 / HBM_PEAK_GBS, unit="GB/s",
                    bytes_per_launch=BYTES_PER_EVAL * n2,
                    note="16 B per pair as above; the pass also reads a 4-byte sample index per pair and the hills' records")
            del d_first2

    nd = None
    if not args.no_nd and rank == 0:
        guard.stage("coordinate_cv")
        nd = coordinate_cv_measure(H, W, tmpdir)

    if rank == 0:
        out = dict(headline)
        out["roofline"] = dict(headline["roofline"])
        out["roofline"]["device_copy_same_traffic_us"] = copy_us
        # the other kernels' rooflines and the metric's second quantity ride INSIDE the roofline object (the record keeps
        # nested objects whole); the top-level copies below stay for readers of earlier rounds' lines
        other = {}
        if roof_w2:
            other["w2_38.8M_pairs_k_pair_forces_fast"] = roof_w2
        if nd:
            for tag in nd:
                other["k2_" + tag] = nd[tag]["roofline"]
        other["hill_adds_strong_scaling"] = headline.get("hill_adds_strong_scaling")
        out["roofline"]["other"] = other
        out.update({
            "evals_only_million_per_s": npairs / t_eval / 1e6,
            "ms_per_step_device_rng": ms_step_device_rng,
            "lj_melt_32k_from_positions": lj,
            "step_modes": step_modes,
            "ms_per_step_hills_log_on": headline["ms_per_step"],
            "ms_per_step_batch_order": step_modes["ms_per_step_batch_order"],
            "forces_only_ms_per_call": t_eval * 1e3,
            "energy_last_step": energy,
            # (batches released by the polled completion word / by the stream-wait fallback, whole run of this object)
            "polled_batches": b.get("polled_batches"),
            "poll_fallbacks": b.get("poll_fallbacks"),
        })
        out.update(extra)
        if roof_w2:
            out["roofline_w2"] = roof_w2
        if nd:
            out["coordinate_cv"] = nd
        out["pcie_inclusive"] = pcie
        guard.stage("cpu_baseline")
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(tmpdir)
            out["cpu_baseline"]["all_cores"] = cpu_baseline_all_cores(tmpdir)
        elif world > 1:
            out["cpu_baseline"] = None
        guard.finish(out)
    else:
        guard.finish(None)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
