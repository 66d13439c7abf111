"""bench.py's output contract (one JSON line; metric / value / roofline / cpu_baseline fields) on a short clip."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines  # exactly ONE line on stdout
    return json.loads(lines[0])


def test_bench_line_contract():
    d = _run("--frames", "12", "--steps", "2", "--warmup", "1")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["unit"] == "frames/s" and d["value"] > 0
    assert abs(d["value"] - d["config"]["encoded_frames_per_step"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and "traffic" in r
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) <= 1e-6 * r["achieved"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] == 1 and c["value"] > 0 and c["unit"] == "frames/s" and c["sample"]
    assert d["hbm_streaming_measured"]["read_only"] > 1000


def test_bench_other_config_and_flags():
    d = _run("--config", "C2-720p-3L-dct8", "--frames", "6", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--wire")
    assert "cpu_baseline" not in d and d["config"]["workload"] == "C2-720p-3L-dct8" and d["value"] > 0
