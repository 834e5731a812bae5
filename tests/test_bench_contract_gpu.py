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
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--cpu-blocks", "32",
                          "--cpu-budget", "6"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["kernel_path"] == 3
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["achieved"] > 0
    # achieved = algorithmic bytes per launch / average launch duration of the dominant kernel
    assert abs(r["achieved"] - r["alg_bytes_per_block"] * r["blocks_per_launch"] / (r["kernel_avg_launch_ms"] * 1e-3) / 1e9) < 1.0
    nb = d["config"]["blocks_per_step_per_gpu"]
    assert nb == 2048 and d["config"]["input_rings"] >= 3      # the headline is cache-cold: no ring can survive in the 256 MiB cache
    assert d["value"] > 1e4 and abs(d["value"] - nb * 32768 / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-2
    assert r["one_ring"] is None and (r["traffic_source"] is None or "pmc" in r["traffic_source"])
    assert d["end_to_end_h2d"]["value"] > 0 and d["end_to_end_h2d"]["blocks_per_call"] == 256
    # one kernel: the dominant kernel is the step, so the contract's frac and the whole-step fraction agree to the launch gaps
    # (the driver's own arguments; a region of 20 launches still carries ~10 us of fixed cost)
    assert 0.75 < r["pipeline_frac"] / r["frac"] <= 1.05
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "Msamples/s" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert c["one_thread"]["cores"] == 1 and 0 < c["one_thread"]["value"] <= c["value"] * 1.5
    assert c["torch_fft"]["value"] > 0 and c["host"]["nproc"] >= c["cores"] and "cpu_model" in c["host"]


def _line(args, env=None, timeout=900):
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout,
                         cwd=ROOT, env=e)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_config4_shape():
    d = _line(["--config", "4", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"])
    assert d["config"]["blocklen"] == 262144 and d["config"]["channels"] == 1024 and d["config"]["kernel_path"] == 2
    assert d["roofline"]["alg_bytes_per_block"] == 2097152 and d["value"] > 1e4


def test_bench_config3_and_5_sinks():
    for cfg in (3, 5):
        d = _line(["--config", str(cfg), "--blocks", "256", "--steps", "3", "--warmup", "1", "--cpu-blocks", "16", "--cpu-budget", "4"])
        assert d["config"]["baseline_config"] == cfg and d["config"]["pdus_per_step"] > 0
        assert d["config"]["extracted_samples_per_step"] > 0
        # B_alg = input bytes + the samples really extracted (data dependent)
        assert d["roofline"]["alg_bytes_per_block"] > 8 * 32768 and d["roofline"]["achieved"] > 0
        assert d["cpu_baseline"]["value"] > 0 and d["value"] > 100.0


def test_bench_sinks_lookahead_form():
    """bench.py --lookahead (round 5): the sink configurations with the producer one batch ahead (FDC_SINKS_LOOKAHEAD, payloads in HBM): the same number of
    PDUs per step as the one-buffer form on the same batch size, a batch of whole rounds of the forward kernel's workgroups, the line says what ran."""
    for cfg in (3, 5):
        la = _line(["--config", str(cfg), "--payload", "device", "--lookahead", "--reserve-cus", "32", "--steps", "4", "--warmup", "2", "--no-cpu-baseline",
                    "--no-end-to-end"])
        c = la["config"]
        nb, wg = c["blocks_per_step_per_gpu"], c["lookahead"]["block_kernel_workgroups"]
        assert c["lookahead"]["reserved_compute_units"] == 32 and nb % wg == 0 and "look-ahead" in c["submission"] and c["payload"] == "device"
        one = _line(["--config", str(cfg), "--payload", "device", "--blocks", str(nb), "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-end-to-end"])
        assert one["config"]["blocks_per_step_per_gpu"] == nb and "lookahead" not in one["config"]
        assert la["config"]["pdus_per_step"] == one["config"]["pdus_per_step"] > 0
        assert la["config"]["extracted_samples_per_step"] == one["config"]["extracted_samples_per_step"]
        assert la["value"] > 100.0


def test_bench_two_ranks_rehearsal_on_one_gpu():
    """bench.py --gpus 2 started plainly: it spawns the two ranks itself; FDC_BENCH_REHEARSE=1 lets both use cuda:0 over gloo,
    so the HIP path runs once per rank (span sharding with halo and global block index)."""
    d = _line(["--gpus", "2", "--steps", "3", "--warmup", "1", "--blocks", "256", "--no-cpu-baseline"],
              env={"FDC_BENCH_REHEARSE": "1"})
    assert d["n_gpus"] == 2 and d["config"]["parallelism"].startswith("block-span sharding x2") and d["value"] > 1e3
