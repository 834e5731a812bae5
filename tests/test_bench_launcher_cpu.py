"""CPU tests (-m "not gpu") of bench.py's own multi-rank launcher: started plainly with --gpus N it must start N ranks
itself (torch.distributed.run children, created before this process touches a GPU), report n_gpus = the ranks that
really ran, and refuse a rank count that does not match.  FDC_BENCH_DRYRUN=1 replaces the GPU work by the timing
plumbing alone (gloo barrier + MAX reduction), so the launcher can be rehearsed on a box without a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, env_extra, timeout=300):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                          timeout=timeout, cwd=ROOT, env=env)


def test_gpus_2_starts_two_ranks_by_itself():
    out = run(["--gpus", "2", "--steps", "2", "--warmup", "1"], {"FDC_BENCH_DRYRUN": "1"})
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and "dry-run" in d["data"]
    assert d["ms_per_step"] >= 20.0                         # MAX over ranks: rank 1 sleeps 20 ms, rank 0 only 10
    # the line shows every rank's own time (a straggler is visible) and what the process group itself reports
    c = d["config"]
    assert c["ranks_seen"] == 2 and c["backend"] == "gloo" and len(c["per_rank_ms"]) == 2
    assert c["per_rank_ms"][1] >= 20.0 > c["per_rank_ms"][0] >= 10.0 and abs(max(c["per_rank_ms"]) - d["ms_per_step"]) < 1e-6


def test_single_rank_plain_start():
    out = run(["--steps", "1", "--warmup", "0"], {"FDC_BENCH_DRYRUN": "1"})
    assert out.returncode == 0 and json.loads(out.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_rank_count_mismatch_is_an_error():
    # a torch.distributed environment with ONE rank while --gpus 2 was asked for: no silent single-GPU number
    out = run(["--gpus", "2", "--steps", "1", "--warmup", "0"],
              {"FDC_BENCH_DRYRUN": "1", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode != 0 and "rank" in (out.stderr + out.stdout)
    out = run(["--gpus", "1", "--steps", "1", "--warmup", "0"],
              {"FDC_BENCH_DRYRUN": "1", "WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "1"})
    assert out.returncode != 0


def test_without_a_gpu_the_real_run_fails_loudly():
    import gr_fdc_amd as G
    if G.lib().fdc_device_count() > 0:
        return
    out = run(["--steps", "1"], {})
    assert out.returncode != 0 and "MI355X" in (out.stderr + out.stdout)
