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
    assert d["hill_adds_strong_scaling"]["value"] > 0
