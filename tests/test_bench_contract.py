"""bench.py's output contract (the driver parses it): ONE JSON line on stdout with the agreed keys, the metric of
BASELINE.json, a roofline object for the dominant kernel.  Runs the real benchmark with a handful of steps."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2",
                          "--no-cpu-baseline", "--no-w2"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "exactly one line on stdout: %r" % lines[:5]
    d = json.loads(lines[0])
    baseline = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "evals/sec" in d["metric"] and "evals/sec" in baseline["metric"]
    assert d["unit"] == "million evals/s" and d["value"] > 0
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2
    assert d["ms_per_step"] > 0 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic"
    assert isinstance(d["config"]["workload"], str) and "model" not in d["config"]
    # value is whole-job pairs per second, consistent with the step time it reports
    pairs = d["config"]["pairs_per_gpu"] * d["n_gpus"]
    assert abs(d["value"] - pairs / (d["ms_per_step"] * 1e-3) / 1e6) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert r["peak"] > 0 and r["achieved"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-9
    assert "traffic" in r and r["launches"] > 0
    # the second quantity of the metric, reported beside it
    assert d["hill_adds_strong_scaling"]["value"] > 0 and d["hill_adds_strong_scaling"]["scaling"] == "strong"
    # both conventions for the fused launch's bytes: SURVEY 8(d)'s 16 B per evaluation is `frac`
    assert r["bytes_per_launch"] == 16 * d["config"]["pairs_per_gpu"]
    assert r["frac_incl_sample_indices"] > r["frac"]
    # the driver's record keeps nested objects whole: the other kernels' rooflines live inside `roofline`
    assert r["other"]["hill_adds_strong_scaling"]["value"] > 0
    for key in ("k2_c2d_2048sq", "k2_c3d_512cube"):
        assert 0 < r["other"][key]["frac"] < 1
    # the default configuration is what is timed: reference fix's order, HILLS log on; the other modes beside it
    assert "reference" in d["config"]["order"] and d["config"]["hills_log"].startswith("on")
    assert d["ms_per_step_hills_log_on"] == d["ms_per_step"]
    sm = d["step_modes"]
    for key in ("ms_per_step_reference_order", "ms_per_step_batch_order", "ms_per_step_reference_order_hills_log_off",
                "ms_per_step_batch_order_hills_log_off"):
        assert sm[key] > 0
    assert d["forces_only_ms_per_call"] > 0
    # BASELINE configs[3] / [4] in the default line: coordinate-CV lookups with their own roofline objects
    for tag, per_atom in (("c2d_2048sq", 156), ("c3d_512cube", 332)):
        c = d["coordinate_cv"][tag]
        rr = c["roofline"]
        assert rr["bytes_per_launch"] == per_atom * c["atoms"] and rr["bound"] == "hbm"
        assert abs(rr["frac"] - rr["achieved"] / rr["peak"]) <= 1e-9 and 0 < rr["frac"] < 1
        assert c["step_ms"] > 0 and c["lookup_replica"]["in_use"] is True
        st = c["step_stats"]
        assert st["min_ms"] <= st["median_ms"] <= st["max_ms"] and st["poll_fallbacks"] >= 0 and st["replica_rebuilds"] >= 0
        assert c["pcie_inclusive_step"]["ms_per_step"] > c["step_ms"]
    assert d["pcie_inclusive"]["ms_per_step"] > d["ms_per_step"]


def test_gpus_flag_without_launcher_never_reports_a_smaller_job():
    """`bench.py --gpus N` with no WORLD_SIZE starts the N ranks itself; where the GPUs are not there it must fail
    loudly (non-zero status, nothing on stdout) -- never run one rank and print n_gpus: 1."""
    import edm_amd.hip as H

    if H.device_count() >= 8:
        pytest.skip("an 8-GPU node would really run the job")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert res.returncode != 0
    assert not res.stdout.strip(), res.stdout[-500:]
    assert "--gpus 8" in res.stderr
