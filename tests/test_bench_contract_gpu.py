"""GPU test (-m gpu): bench.py prints exactly one JSON line that carries the driver's contract fields, the roofline
object (dominant kernel timed by HIP events inside the run) and, at N = 1, the CPU baseline."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--cpu-blocks", "32"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["kernel_path"] == 2
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["achieved"] > 0
    # achieved = algorithmic bytes per launch / average launch duration of the dominant kernel
    assert abs(r["achieved"] - r["alg_bytes_per_block"] * r["blocks_per_launch"] / (r["kernel_avg_launch_ms"] * 1e-3) / 1e9) < 1.0
    assert d["value"] > 1e4 and abs(d["value"] - 1024 * 32768 / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-2
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "Msamples/s" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
