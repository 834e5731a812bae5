#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X frequency-domain channelizer.

Metric (BASELINE.json): complex Msamples/s IN on the 65536-pt FFT / 50 % overlap-save / 256-channel
channelizer (configs[1]); roofline = algorithmic HBM bytes (B_alg = 8*H + 8*sum(lout) =
524288 B per input block) over the dominant kernel's measured launch time, against 8 TB/s.

A "step" is one pass of the hot path (overlap-save gather -> forward FFT -> fused per-channel
slice/window/IFFT/discard) over one batch of device-resident synthetic multicarrier input.
  python bench.py [--gpus N --steps K --warmup W] ; for N>1 launched by torch.distributed.run,
one rank per GPU, each rank owning an independent contiguous span of blocks (halo + global block index:
SURVEY.md §8e) — no data-path collective; weak scaling.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--blocks", type=int, default=1024, help="input blocks per step per GPU")
    ap.add_argument("--blocklen", type=int, default=65536)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--relinvovl", type=int, default=2)
    ap.add_argument("--chunk", type=int, default=0, help="blocks per launch group (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="diagnostics: no HIP events inside the timed region (no roofline numbers)")
    ap.add_argument("--timing-stride", type=int, default=4,
                    help="HIP events around the kernels of every k-th launch group of the timed region (their packets cost "
                         "7-17 us per group; 1 = every group)")
    ap.add_argument("--cpu-blocks", type=int, default=0, help="blocks in the CPU baseline sample (0 = auto)")
    ap.add_argument("--check", action="store_true", help="verify a few blocks against the oracle first")
    return ap.parse_args()


def synth_input(torch, dev, N, R, C, nblocks, first_block, seed):
    """Device-resident synthetic multicarrier ring: N/R halo samples + nblocks*H new samples (SURVEY §8d cfg2:
    one carrier per channel, random complex symbols at 0.6x the channel bandwidth, noise at -30 dB)."""
    H = N - N // R
    ovl = N // R
    n0 = first_block * H - ovl
    total = ovl + nblocks * H
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    x = torch.randn(total, 2, device=dev, generator=g, dtype=torch.float32) * (10 ** (-30 / 20) / 2 ** 0.5)
    x = torch.view_as_complex(x).contiguous()
    sps = max(2, int(round(C / 0.6)))            # samples per symbol at 0.6x the channel bandwidth
    nsym = total // sps + 2
    step = 1 << 20
    cgrp = 32
    for c0 in range(0, C, cgrp):
        cc = min(cgrp, C - c0)
        sym = (torch.randint(0, 2, (cc, nsym, 2), device=dev, generator=g, dtype=torch.int32).float() * 2 - 1) * (0.5 ** 0.5)
        sym = torch.view_as_complex(sym.contiguous())
        fc = (torch.arange(c0, c0 + cc, device=dev, dtype=torch.float64) + 0.5) / C - 0.5     # cycles/sample
        for s0 in range(0, total, step):
            s1 = min(total, s0 + step)
            n = torch.arange(n0 + s0, n0 + s1, device=dev, dtype=torch.float64)
            ph = (fc[:, None] * n[None, :]) % 1.0
            car = torch.polar(torch.ones_like(ph, dtype=torch.float32), (2 * torch.pi * ph).float())
            idx = (torch.arange(s0, s1, device=dev) // sps)
            x[s0:s1] += (sym[:, idx] * car).sum(0)
    if first_block == 0:
        x[:ovl] = 0                               # stream start: zero history (lib/overlap_save_impl.cc:52)
    return x


def cpu_baseline(N, R, plan, nthreads, blocks, budget_s=12.0):
    """Times the oracle (float32 arithmetic, OpenMP over blocks) on the host cores: kind = "port".
    Bounded sample: passes over the same `blocks`-block buffer until about budget_s seconds of wall time."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O          # cpu_baseline leg: the oracle is what is measured here, by design
    H = N - N // R
    rng = np.random.default_rng(2025)
    x = (rng.standard_normal(blocks * H) + 1j * rng.standard_normal(blocks * H)).astype(np.complex64)
    O.channelizer(N, R, 1, plan, x[:nthreads * H], use_float=True, nthreads=nthreads)     # warm-up (plans, pages)
    reps, t0 = 0, time.perf_counter()
    while True:
        O.channelizer(N, R, 1, plan, x, use_float=True, nthreads=nthreads)
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or reps >= 1000:
            break
    return reps * blocks * H / dt / 1e6, dt, reps


def main():
    a = parse()
    import torch                      # first: my library then binds to the same libamdhip64 torch loaded
    import numpy as np
    import gr_fdc_amd as G

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # rehearsal on a one-GPU box (FDC_BENCH_REHEARSE=1): all ranks share cuda:0 and talk over gloo; the driver's real
    # multi-GPU run is one rank per GPU over RCCL
    rehearse = os.environ.get("FDC_BENCH_REHEARSE") == "1"
    if rehearse:
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    N, R, C, nb = a.blocklen, a.relinvovl, a.channels, a.blocks
    H = N - N // R
    # channel plan through the reference's own parameter derivation (py:322-345): tiles the spectrum
    params = [G.get_opt_channelparams(N, R, ((c + 0.5) / C - 0.5 + 0.5) % 1.0, 0.8 / C) for c in range(C)]
    plan = [(f, l, p, s) for (f, l, _lo, p, s) in params]
    sum_lout = sum(lo for (_f, _l, lo, _p, _s) in params)
    b_alg = 8 * H + 8 * sum_lout                       # SURVEY.md §8d, bytes per input block

    pipe = G.Pipeline(N, R, plan, windowtype=1, max_blocks=nb, device_id=local, chunk_blocks=a.chunk)
    first_block, _n = G.span_for_rank(world * nb, rank, world)   # contiguous span per rank (§8e); weak scaling
    x = synth_input(torch, dev, N, R, C, nb, first_block, 2025 + rank)
    out = torch.empty(pipe.output_samples(nb), dtype=torch.complex64, device=dev)
    torch.cuda.synchronize()

    if a.check and rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as O
        k = 4
        pipe.process_device(x.data_ptr(), first_block, nb, out.data_ptr())
        pipe.synchronize()
        xs = x[:N // R + k * H].cpu().numpy()
        ref, _ = O.channelizer(N, R, 1, plan, xs[N // R:], prefix=xs[:N // R], first_block=first_block, nthreads=8)
        worst = 0.0
        for c in range(C):
            o = out[pipe.channel_offset(c, nb):pipe.channel_offset(c, nb) + k * params[c][2]].cpu().numpy()
            worst = max(worst, float(np.abs(o - ref[c]).max() / np.abs(ref[c]).max()))
        print("check: max rel err vs oracle over %d blocks x %d channels = %.3g" % (k, C, worst), file=sys.stderr)
        assert worst <= 1e-5

    def step():
        pipe.process_device(x.data_ptr(), first_block, nb, out.data_ptr())

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    pipe.enable_timing(0 if a.no_kernel_timing else max(1, a.timing_stride))
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    # per-kernel HIP-event durations, summed over every launch of the timed region (the events sit on the
    # pipeline's own stream, the one the kernels are launched on); read out after the region is closed
    last = pipe.last_kernel_ms()
    pipe.enable_timing(False)
    if dist is not None:
        t = torch.tensor([dt], device="cpu" if rehearse else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    msps = world * nb * H * a.steps / dt / 1e6
    chunk = pipe.chunk_blocks()
    ngroups = max(1, int(last[3]))              # launch groups that carried events (every timing_stride-th of the region)
    nlaunch = (nb + chunk - 1) // chunk         # launch groups per step
    path = pipe.path()
    names = ["block_kernel(colFFT+window+IFFT+slotFFT)", "unused", "unused2"] if path == 3 else \
            ["poly_stage1(colFFT+window+IFFT)", "poly_stage2(slotFFT)", "unused"] if path == 2 else \
            ["fft_pass_a", "fft_pass_b", "channels"]
    dom = max(range(3), key=lambda i: last[i])
    dom_avg_ms = last[dom] / ngroups            # average duration of ONE launch of the dominant kernel
    blocks_per_launch = nb / nlaunch            # units one launch processes
    achieved = b_alg * blocks_per_launch / (dom_avg_ms * 1e-3) / 1e9 if dom_avg_ms > 0 else 0.0
    # HBM bytes per launch of the dominant kernel from rocprofv3 PMC passes of this same command
    # (profiles/pmc_run.sh -> profiles/pmc_traffic.json; FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM)
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            pt = json.load(fh)
        ent = pt.get(names[dom])
        if ent and ent.get("blocks_per_launch") == blocks_per_launch and ent.get("blocklen") == N:
            traffic = ent["hbm_bytes_per_launch"]
    except (OSError, ValueError):
        pass
    res = {
        "metric": "complex Msamples/s in, 64k-FFT/256-ch overlap-save; achieved HBM GB/s vs peak",
        "value": round(msps, 3), "unit": "Msamples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: %d-pt FFT, 1/%d overlap-save, %d fixed channels (l=%d, lout=%d), "
                               "%d blocks/step/GPU" % ("configs[1]" if (N, R, C) == (65536, 2, 256) else
                                                       "configs[3] per-GPU shape" if (N, R, C) == (262144, 2, 1024) else
                                                       "non-default shape", N, R, C, params[0][1], params[0][2], nb),
                   "blocklen": N, "relinvovl": R, "channels": C, "blocks_per_step_per_gpu": nb,
                   "chunk_blocks": chunk, "kernel_path": path, "parallelism": "block-span sharding x%d, no collective" % world},
        "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "kernel_ms_per_step": {n: round(v / ngroups * nlaunch, 4) for n, v in zip(names, last)},
                     "kernel_avg_launch_ms": round(dom_avg_ms, 5), "blocks_per_launch": blocks_per_launch,
                     "launches_per_step": nlaunch, "timed_launches": ngroups, "timing_stride": max(1, a.timing_stride),
                     "alg_bytes_per_block": b_alg,
                     "pipeline_achieved": round(b_alg * nb * a.steps / dt / 1e9, 2),
                     "pipeline_frac": round(b_alg * nb * a.steps / dt / 1e9 / HBM_PEAK_GBS, 4),
                     # SURVEY.md §8d also asks for the fraction of the achievable float4-copy rate (6.3 TB/s per the guide)
                     "frac_of_achievable_6300": round(achieved / 6300.0, 4),
                     "pipeline_frac_of_achievable_6300": round(b_alg * nb * a.steps / dt / 1e9 / 6300.0, 4)},
    }
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            ncores = len(os.sched_getaffinity(0))
        except AttributeError:
            ncores = os.cpu_count() or 1
        nthreads = min(ncores, 16)          # a one-GPU box's CPU share
        cb = a.cpu_blocks or max(nthreads * 4, 128)
        v, secs, reps = cpu_baseline(N, R, plan, nthreads, cb)
        import ctypes.util
        have = [n for n in ("fftw3f", "volk") if ctypes.util.find_library(n)]      # SURVEY.md §8d start-up probe
        res["cpu_baseline"] = {"value": round(v, 3), "unit": "Msamples/s", "cores": nthreads, "kind": "port",
                               "sample": "%d passes over %d blocks of the same workload (N=%d, %d channels), "
                                         "float32 oracle port, OpenMP over blocks, %.1f s wall; host libraries of the "
                                         "reference's own CPU path found: %s"
                                         % (reps, cb, N, C, secs, ", ".join(have) if have else "none (no FFTW3f, no VOLK)")}
    if rank == 0:
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
