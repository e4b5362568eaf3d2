"""bench.py keeps the driver's contract: one JSON line with the agreed keys (GPU), and no CPU
fallback (without a device it stops with a message instead of measuring anything)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=600,
                          env={**os.environ, **(env or {})})


def test_bench_refuses_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    r = _run(["--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and "HIP device" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_bench_json_contract():
    r = _run(["--workload", "config2", "--steps", "4", "--warmup", "2", "--cpu-steps", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["achieved"] > 0 and "traffic" in rf
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == d["unit"] and cb["sample"]
    assert d["value"] > 1e8 and abs(d["value"] - 262144 * 4 / (d["ms_per_step"] * 4e-3)) / d["value"] < 1e-6
