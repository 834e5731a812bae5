#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X frequency-domain channelizer.

Metric (BASELINE.json): complex Msamples/s IN on the 65536-pt FFT / 50 % overlap-save / 256-channel
channelizer (configs[1]); roofline = algorithmic HBM bytes (B_alg = 8*H + 8*sum(lout) =
524288 B per input block) over the dominant kernel's measured launch time, against 8 TB/s.

A "step" is one pass of the hot path (overlap-save gather -> forward FFT -> fused per-channel
slice/window/IFFT/discard) over one batch of device-resident synthetic multicarrier input.
  python bench.py [--gpus N --steps K --warmup W] [--config 2|3|4|5]

--config picks the BASELINE.json workload (numbered as SURVEY.md §8d does, cfgK = configs[K-1]):
  1            configs[0]: the example flowgraph's plan (4096-pt FFT, 4 channels of l = 256/512/1024/512), a mixed plan on
               the generic kernels; 16384 blocks per step (the same sample count as the headline)
  2 (default)  configs[1]: 65536-pt FFT, 256 fixed channels                        <- the headline line
  3            configs[2]: the same 256 channels as PowerActivationChannel sinks, bursty carriers
  4            configs[3]'s per-GPU shape: 262144-pt FFT, 1024 fixed channels, 256 blocks per step
  5            configs[4]: activity_detection_channelizer_vcm, two segments, 24 bursty carriers

--gpus N > 1: one rank per GPU, each rank an independent contiguous span of blocks (halo + global block index:
SURVEY.md §8e), no data-path collective, weak scaling.  Under torch.distributed.run the ranks are used as given;
started plainly, bench.py starts the N ranks itself — child processes, created before anything in this process
touches the GPU — and exits with their status.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")     # CPU baseline legs: one leg's idle OpenMP team must not spin into the next

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
METRIC = "complex Msamples/s in, 64k-FFT/256-ch overlap-save; achieved HBM GB/s vs peak"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle-ms", type=float, default=150.0, help="untimed running-in before the W warm-up steps: the device leaves its idle "
                                                                   "clocks only under load (0 = none)")
    ap.add_argument("--config", type=int, default=2, choices=(1, 2, 3, 4, 5), help="BASELINE workload, see the module text")
    ap.add_argument("--blocks", type=int, default=0, help="input blocks per step per GPU (0 = the workload's default)")
    ap.add_argument("--blocklen", type=int, default=0)
    ap.add_argument("--channels", type=int, default=0)
    ap.add_argument("--relinvovl", type=int, default=2)
    ap.add_argument("--chunk", type=int, default=0, help="blocks per launch group (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="diagnostics: no HIP events inside the timed region (no roofline numbers)")
    ap.add_argument("--timing-stride", type=int, default=4,
                    help="HIP events around the kernels of every k-th launch group of the timed region (their packets cost "
                         "7-17 us per group; 1 = every group)")
    ap.add_argument("--cpu-blocks", type=int, default=0, help="blocks in the CPU baseline sample (0 = auto)")
    ap.add_argument("--cpu-budget", type=float, default=15.0, help="seconds of CPU work for the whole cpu_baseline object")
    ap.add_argument("--check", action="store_true", help="verify a few blocks against the oracle first")
    ap.add_argument("--offset", type=int, default=0, help="diagnostics (configs 2/4): every channel r bins higher (the last one "
                                                             "dropped): a tiling that does not start at bin 0")
    ap.add_argument("--force-path", choices=("no-block", "no-poly", "generic", "full-spectrum", "wide-uniform", "no-fused"), default=None,
                    help="diagnostics: fdc_pipeline_cfg.flags FDC_PIPE_NO_BLOCK / NO_POLY / FORCE_GENERIC (the slower forms of the path)")
    ap.add_argument("--block-hints", type=int, default=None, metavar="H",
                    help="diagnostics: memory hints of the block kernels (bit 0: streamed output stores, bit 1: streamed input loads; default 1)")
    ap.add_argument("--input-rings", type=int, default=3,
                    help="distinct device-resident input rings the steps rotate over (configs 1/2/4): with 3 rings of 537 MB (2048 blocks of "
                         "262144 B) no input byte of a step can still sit in the 256 MiB memory-side cache when its ring comes round again "
                         "(1 = the round-1/2 form with 1024 blocks, where it can)")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle check of the last timed launch's output (the `verified` object)")
    ap.add_argument("--verify-blocks", type=int, default=8, help="blocks of the last timed launch checked against the oracle (first, last, random)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the host-buffer legs (end_to_end_h2d, end_to_end_h2d_group)")
    ap.add_argument("--one-ring-leg", action="store_true",
                    help="config 2: after the timed region, the same steps on ONE 268 MB ring (the round-1/2 workload, whose input stays in "
                         "the memory-side cache) reported as roofline.one_ring; off by default so that the rocprofv3 average of the dominant "
                         "kernel over the default command is the average of the timed launches")
    ap.add_argument("--payload", choices=("host", "device"), default="host",
                    help="configs 3/5: PDU payloads copied to pinned host memory (default; PCIe-bound) or left in HBM (fdc_pdu.samples "
                         "are device pointers)")
    ap.add_argument("--sink-engine", choices=("device", "host"), default="device",
                    help="configs 3/5: where the blocks' work() loops run (FDC_SINKS_HOST_DECISIONS = the round-2 form)")
    ap.add_argument("--noise-input", action="store_true", help="diagnostics (configs 1/2/4): the ring is complex noise without carriers — kernel times and counters do not "
                    "depend on the data; for rocprofv3 --pmc runs of plans whose synthesis kernels the profiler trips over (1024 carriers: it segfaults)")
    ap.add_argument("--lookahead", action="store_true", help="configs 3/5: the bank with two spectrum buffers (FDC_SINKS_LOOKAHEAD): the forward transform and the "
                    "power cells of batch n + 1 run on the bank's fill stream beside the decision kernels of batch n; the block kernels leave "
                    "--reserve-cus compute units to them and the batch is a multiple of their workgroups")
    ap.add_argument("--no-fused-cells", action="store_true", help="configs 3 / 5: power cells by a pass over the spectrum (k_cell_power, the round-5 form) instead of "
                                                                   "from the group sums of the forward kernel's epilogue")
    ap.add_argument("--reserve-cus", type=int, default=32, help="--lookahead: compute units the persistent forward-transform kernel leaves free")
    ap.add_argument("--sync-sinks", action="store_true", help="configs 3/5: fdc_sinks_work_device per step instead of the two-deep "
                                                               "fdc_sinks_submit_device")
    ap.add_argument("--mixed", action="store_true", help="diagnostics (config 2): the same centres with bandwidths cycling through "
                                                         "0.8/C, 0.4/C, 0.8/C, 1.6/C -> a mixed-width plan (spectrum path)")
    ap.add_argument("--two-widths", action="store_true", help="diagnostics (config 2): the lower half of the band as 128 channels of 256 bins, the upper half as "
                                                              "64 channels of 512 bins: two BANKS of different widths, one block-kernel launch each (round 5)")
    ap.add_argument("--width", type=int, default=0, metavar="L", help="diagnostics (config 2): a uniform bank of channels L bins wide (N / L channels "
                                                                      "on the L-bin grid) instead of 256: the generic-width two-launch path")
    ap.add_argument("--gapless", type=int, default=0, metavar="C", help="diagnostics (config 2): C channels of 1/C of the band each (no gaps): slices of 2 N / C bins "
                                                                          "overlapping by half, a quarter of a channel off their grid (with --centred: on it and half off it)")
    ap.add_argument("--centred", action="store_true", help="the channels centred on k/C instead of (k + 1/2)/C (a bank half a channel off "
                                                          "the grid plus the wrapped channel 0 on it: two launches of the width's kernel)")
    ap.add_argument("--extra", type=int, default=0, metavar="K", help="diagnostics (config 2): K more channels of widths 512 / 128 / 1024 at odd bins beside "
                                                                      "the 256-channel bank: an ALMOST uniform plan (split: the bank on the one-kernel "
                                                                      "form, the K others on a partial spectrum; kernel_path 4)")
    ap.add_argument("--sparse", type=int, default=0, metavar="K", help="diagnostics (config 2): K channels (widths 256 ... 2048 in turn) spread "
                                                                       "over the band instead of a plan that tiles it (spectrum path)")
    ap.add_argument("--sparse-widths", default="256,512,1024,2048", help="--sparse: the channel widths in turn (bins, powers of two)")
    a = ap.parse_args()
    if a.sparse and a.config != 2:
        ap.error("--sparse applies to --config 2 only")
    if a.offset and a.config not in (2, 4):
        ap.error("--offset applies to --config 2 and 4 only")
    if a.mixed and a.config != 2:
        ap.error("--mixed applies to --config 2 only")
    if a.config == 1:      # the reference's own example flowgraph (plumbing case): 4096-pt FFT, its 4 channels of mixed width
        a.blocklen, a.channels, a.blocks = a.blocklen or 4096, 4, a.blocks or 16384
    elif a.config == 4:
        a.blocklen, a.channels, a.blocks = a.blocklen or 262144, a.channels or 1024, a.blocks or 256
    else:
        a.blocklen, a.channels, a.blocks = a.blocklen or 65536, a.channels or 256, a.blocks or (2048 if a.config == 2 else 1024)
    return a


def launch_ranks(a):
    """--gpus N without a torch.distributed environment: start the N ranks as children and pass their status on.
    Nothing in THIS process has touched the GPU (torch is not even imported yet)."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


def gpu_numa_cpus(ordinal):
    """CPUs of the NUMA node GPU `ordinal` hangs off, from sysfs alone (nothing here touches the GPU: it runs before the first HIP call).
    HIP lists the KFD topology's GPU nodes in order (ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES given as plain indices are applied);
    node properties carry the PCI domain and location_id = bus << 8 | devfn.  None when anything is missing (containers, one-node hosts)."""
    try:
        gpus = []
        base = "/sys/class/kfd/kfd/topology/nodes"
        for nd in sorted(os.listdir(base), key=int):
            props = dict(ln.split()[:2] for ln in open(os.path.join(base, nd, "properties")) if len(ln.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                loc, dom = int(props["location_id"]), int(props.get("domain", "0"))
                gpus.append("%04x:%02x:%02x.%x" % (dom, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 7))
        for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
            v = os.environ.get(var)
            if v:
                gpus = [gpus[int(i)] for i in v.split(",")]
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % gpus[ordinal]).read())
        if node < 0:
            return None
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus |= set(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        return (node, cpus) if cpus else None
    except (OSError, ValueError, KeyError, IndexError):
        return None


def synth_input(torch, dev, N, R, C, nblocks, first_block, seed, carriers=True):
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
    for c0 in range(0, C if carriers else 0, cgrp):
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


def synth_bursty(torch, dev, N, R, carriers, nblocks, seed, floor_db=-30.0):
    """Device-resident ring for the stateful sinks: every carrier (centre in cycles/sample,
    relative width) is on/off-keyed in bursts of 8-64 blocks at 50 % duty, independently; the noise floor sits
    `floor_db` below a carrier (on/off ratio 30 dB, far beyond the 6 / 10 dB thresholds)."""
    import numpy as np
    H = N - N // R
    ovl = N // R
    total = ovl + nblocks * H
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    x = torch.randn(total, 2, device=dev, generator=g, dtype=torch.float32) * (10 ** (floor_db / 20) / 2 ** 0.5)
    x = torch.view_as_complex(x).contiguous()
    rng = np.random.default_rng(seed)
    step = 1 << 19
    cgrp = 32
    for c0 in range(0, len(carriers), cgrp):
        grp = carriers[c0:c0 + cgrp]
        gate = np.zeros((len(grp), nblocks + 1), np.float32)
        for i in range(len(grp)):
            m, on = int(rng.integers(0, 64)) - 32, bool(rng.integers(0, 2))
            while m < nblocks + 1:
                ln = int(rng.integers(8, 65))
                if on:
                    gate[i, max(m, 0):m + ln] = 1.0
                m += ln
                on = not on
        gate_d = torch.from_numpy(gate).to(dev)
        fc = torch.tensor([c[0] for c in grp], device=dev, dtype=torch.float64)
        sps = torch.tensor([max(2, int(round(1.0 / (0.6 * c[1])))) for c in grp], device=dev)
        sym = (torch.randint(0, 2, (len(grp), 2, 4096), device=dev, generator=g, dtype=torch.int32).float() * 2 - 1) * (0.5 ** 0.5)
        sym = torch.complex(sym[:, 0], sym[:, 1])
        for s0 in range(0, total, step):
            s1 = min(total, s0 + step)
            n = torch.arange(s0, s1, device=dev, dtype=torch.float64)
            ph = (fc[:, None] * n[None, :]) % 1.0
            car = torch.polar(torch.ones_like(ph, dtype=torch.float32), (2 * torch.pi * ph).float())
            nn = torch.arange(s0, s1, device=dev)
            sidx = (nn[None, :] // sps[:, None]) % 4096
            blk = torch.clamp((nn - ovl).div(H, rounding_mode="floor"), 0, nblocks)
            x[s0:s1] += (torch.gather(sym, 1, sidx) * car * gate_d[:, blk]).sum(0)
    x[:ovl] = 0
    return x


def host_info():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        share = len(os.sched_getaffinity(0))
    except AttributeError:
        share = os.cpu_count() or 1
    # a one-GPU box is given 16 host cores' worth of the machine whatever the affinity mask says: the thread pools of the
    # CPU baseline are sized to that share
    return model, os.cpu_count() or 1, min(share, 16)


def cpu_baseline(N, R, plan, blocks, budget_s):
    """The reference's throughput chain on the host cores, same workload shape, a bounded
    sample: (a) the oracle's float32 port at ONE thread, (b) the same on every core of this process's CPU share
    (OpenMP over blocks) — the reported `value`, (c) the chain written with torch.fft on the CPU, a third-party
    transform as a sanity point.  FFTW3f / VOLK (the reference's own libraries) are probed for and reported."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O          # cpu_baseline leg: the oracle is what is measured here, by design
    H = N - N // R
    model, nproc, share = host_info()
    rng = np.random.default_rng(2025)
    x = (rng.standard_normal(blocks * H) + 1j * rng.standard_normal(blocks * H)).astype(np.complex64)

    def timed(fn, budget, unit_blocks):
        fn()                                     # warm-up (plans, pages)
        reps, t0 = 0, time.perf_counter()
        while True:
            fn()
            reps += 1
            dt = time.perf_counter() - t0
            if dt >= budget or reps >= 1000:
                break
        return reps * unit_blocks * H / dt / 1e6, dt, reps

    b1 = max(2, min(blocks, max(4, blocks // share)))
    v1, s1, r1 = timed(lambda: O.channelizer(N, R, 1, plan, x[:b1 * H], use_float=True, nthreads=1), budget_s * 0.2, b1)
    va, sa, ra = timed(lambda: O.channelizer(N, R, 1, plan, x, use_float=True, nthreads=share), budget_s * 0.4, blocks)
    out = {"value": round(va, 3), "unit": "Msamples/s", "cores": share, "kind": "port",
           "sample": "%d passes over %d blocks of the same workload (N=%d, %d channels), float32 oracle port, OpenMP over "
                     "blocks on %d threads, %.1f s wall" % (ra, blocks, N, len(plan), share, sa),
           "one_thread": {"value": round(v1, 3), "cores": 1, "sample": "%d passes over %d blocks, %.1f s" % (r1, b1, s1)},
           "host": {"cpu_model": model, "nproc": nproc, "cpu_share_of_this_process": share}}
    # torch.fft on the CPU: overlap-save blocks -> fft -> fftshift, 1/N -> slices * window -> ifftshift -> ifft -> discard -> *l
    ls = sorted(set(c[1] for c in plan))
    if len(ls) == 1:
        l = ls[0]
        f = torch.tensor([c[0] for c in plan])
        wins = torch.from_numpy(np.stack([O.window(1, l, np.float32(c[2]), np.float32(c[3]), R)[0] for c in plan]))
        idx = (f[:, None] + torch.arange(l)[None, :]).reshape(-1)
        bt = max(2, min(blocks, 16))
        xt = torch.from_numpy(np.concatenate([np.zeros(N // R, np.complex64), x[:bt * H]]))
        torch.set_num_threads(share)
        try:
            torch.set_num_interop_threads(1)
        except RuntimeError:
            pass                                  # already started: stays as it is

        def chain():
            blk = xt.unfold(0, N, H)[:bt]
            spec = torch.fft.fftshift(torch.fft.fft(blk, dim=-1), dim=-1) / N
            sl = spec[:, idx].reshape(bt, len(plan), l) * wins[None]
            y = torch.fft.ifft(torch.fft.ifftshift(sl, dim=-1), dim=-1) * (l * l)
            return y[:, :, l // R:]
        tv, st, rt = timed(chain, budget_s * 0.2, bt)
        out["torch_fft"] = {"value": round(tv, 3), "cores": share,
                            "sample": "%d passes over %d blocks, torch.fft (CPU) for both transforms, %.1f s" % (rt, bt, st)}
    # The reference's own arithmetic libraries, if this host has them: the chain with FFTW3f (4
    # threads on the forward transform, py:206) and VOLK behind the reference's stage boundaries (oracle/ref_equiv.c, dlopen)
    why = O.refequiv_probe()
    if why is None:
        try:
            vr, pr, o0 = O.refequiv_run(N, R, 1, plan, x, 4, budget_s * 0.2)
            ref0, _ = O.channelizer(N, R, 1, plan[:1], x, nthreads=share)
            dev = float(np.abs(o0 - ref0[0]).max() / max(1e-30, np.abs(ref0[0]).max()))
            out["reference_equivalent"] = {"value": round(vr, 3), "unit": "Msamples/s", "cores": share, "kind": "reference-equivalent",
                                           "sample": "%d passes over %d blocks: FFTW3f (forward transform on 4 threads) + VOLK behind the "
                                                     "reference's stage boundaries, channel branches over %d OpenMP threads" % (pr, blocks, share),
                                           "max_rel_dev_from_oracle_channel0": dev}
        except Exception as e:      # noqa: BLE001  (a baseline leg must not take the bench line down)
            out["reference_equivalent"] = {"error": str(e)}
    out["reference_libraries_found"] = "libfftw3f + libvolk" if why is None else \
        "%s: the reference's own CPU path cannot be timed on this box, the port is the reported baseline" % why
    return out


def cpu_baseline_sinks(a, N, R, C, x, nb, segments):
    """cfg3 / cfg5 on the host: the oracle's forward transform (float32 port, OpenMP over blocks) followed by the oracle's
    restatement of the sink block's work() loop, on a bounded prefix of the same input."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O          # cpu_baseline leg
    H = N - N // R
    model, nproc, share = host_info()
    blocks = min(a.cpu_blocks or max(share * 2, 32), nb)
    xs = x[:N // R + blocks * H].cpu().numpy()
    t0 = time.perf_counter()
    reps, npdu = 0, 0
    while True:
        _o, spec = O.channelizer(N, R, 1, [], xs[N // R:], prefix=xs[:N // R], want_spectrum=True, use_float=True, nthreads=share)
        if a.config == 3:
            npdu = 0
            for c in range(C):
                npdu += len(O.PowerActivationChannel(N, ((c + 0.5) / C) % 1.0, 0.8 / C, R, 6.0, 128, 1, c).work(spec))
        else:
            npdu = len(O.ActivityDetectionVcm(N, segments, 10.0, R, 128, 0.005, 1, 0.2).work(spec))
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= a.cpu_budget * 0.8 or reps >= 100:
            break
    return {"value": round(reps * blocks * H / dt / 1e6, 3), "unit": "Msamples/s", "cores": share, "kind": "port",
            "sample": "%d passes over the first %d blocks of the same input: oracle forward transform (float32 port, %d OpenMP "
                      "threads) + the oracle's sink work() loops (1 thread, like the reference block), %d PDUs per pass, %.1f s"
                      % (reps, blocks, share, npdu, dt),
            "host": {"cpu_model": model, "nproc": nproc, "cpu_share_of_this_process": share}}


def verify_last_launch(torch, np, a, pipe, ring, out, plan, params, N, R, nb, first_block, rank):
    """Blocks of the output buffer the last timed launch wrote, against the oracle run on the input that launch read."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O          # the checker, after the timed region
    H = N - N // R
    k = max(2, min(a.verify_blocks, nb))
    rng = np.random.default_rng(4242 + rank)
    blocks = sorted(set([0, nb - 1] + [int(v) for v in rng.integers(0, nb, size=max(0, k - 2))]))
    chans = list(range(0, len(plan), 16)) or [0]
    if len(plan) - 1 not in chans:
        chans.append(len(plan) - 1)
    sub = [plan[c] for c in chans]
    worst_l2, worst_mx = 0.0, 0.0
    for m in blocks:
        xs = ring[m * H:m * H + N].cpu().numpy()                     # the block as overlap-save forms it: N/R old + H new samples
        ref, _ = O.channelizer(N, R, 1, sub, xs[N // R:], prefix=xs[:N // R], first_block=first_block + m, nthreads=4)
        for c, r in zip(chans, ref):
            lo = params[c][2]
            o0 = pipe.channel_offset(c, nb) + m * lo
            got = out[o0:o0 + lo].cpu().numpy()
            den = float(np.abs(r).max()) or 1.0
            worst_mx = max(worst_mx, float(np.abs(got - r).max()) / den)
            worst_l2 = max(worst_l2, float(np.linalg.norm(got - r) / (np.linalg.norm(r) or 1.0)))
    err = max(worst_l2, worst_mx)
    return {"blocks": len(blocks), "block_indices": blocks, "channels": len(chans), "max_rel_err": err, "tolerance": 1e-5,
            "what": "output of the last timed launch vs the oracle on the ring it read (max over rel. L2 and max-abs/max per block and channel)"}


def host_entry_leg(G, np, N, R, plan, sum_lout, devices, blocks_per_member):
    """fdc_pipeline_work (one device) or fdc_pipeline_group_work (several) on pinned caller buffers: Msamples/s in, PCIe included."""
    import ctypes as ct
    H = N - N // R
    k = len(devices)
    nbh = blocks_per_member * k
    if k == 1:
        ph = G.Pipeline(N, R, plan, windowtype=1, max_blocks=nbh, device_id=devices[0])
    else:
        ph = G.PipelineGroup(N, R, plan, devices, windowtype=1, max_blocks=nbh)
    rng_h = np.random.default_rng(7)
    xh = np.empty(nbh * H, np.complex64)
    for b0 in range(0, nbh, 64):                                        # generated in pieces: the complex128 temporaries stay small
        n = min(64, nbh - b0) * H
        xh[b0 * H:b0 * H + n] = (rng_h.standard_normal(n) + 1j * rng_h.standard_normal(n)).astype(np.complex64)
    pool = np.zeros(nbh * sum_lout, np.complex64)
    ptrs, off = (ct.c_void_p * len(plan))(), 0
    for c, lo in enumerate(ph.lout):
        ptrs[c] = pool.ctypes.data + 8 * off
        off += nbh * lo
    G.register_host(xh); G.register_host(pool)
    try:
        for _ in range(2):
            ph.work_raw(xh.ctypes.data, nbh, ptrs)
        th = time.perf_counter()
        reps_h = 8
        for _ in range(reps_h):
            ph.work_raw(xh.ctypes.data, nbh, ptrs)
        dth = (time.perf_counter() - th) / reps_h
        res = {"value": round(nbh * H / dth / 1e6, 3), "unit": "Msamples/s", "blocks_per_call": nbh, "ms_per_call": round(dth * 1e3, 3),
               "devices": list(devices),
               "entry": ("fdc_pipeline_work" if k == 1 else "fdc_pipeline_group_work: one call cut into %d spans, one per member" % k) +
                        " (H2D of the input + kernels + D2H of every channel output per call; buffers pinned with fdc_host_register)"}
        if k > 1:
            res["spans_of_a_call"] = ph.last_spans()
    finally:
        G.unregister_host(xh); G.unregister_host(pool)
        ph.close()
    return res


def sinks_host_entry_leg(G, np, _lib, N, R, local, x, nb, make_bank, per_call, pipelined):
    """The sink configurations through fdc_pipeline_work_sinks from pinned HOST samples: `per_call` items per call over the bench's own
    bursty stream (laps of its nb blocks), PDUs and payloads to host memory and counted.  pipelined: a look-ahead bank (front end of
    call n beside the sinks of call n - 1, PDUs two calls later); otherwise the serial form (front end, then the sinks, inside the call)."""
    import ctypes as ct
    H = N - N // R
    per_call = min(per_call, nb)
    calls_per_lap = nb // per_call
    xh = x[N // R:N // R + calls_per_lap * per_call * H].cpu().numpy()
    pipe = G.Pipeline(N, R, [], windowtype=1, max_blocks=per_call, device_id=local, keep_spectrum=True)
    bank = make_bank(per_call, pipelined, False)
    none = (ct.c_void_p * 1)()
    G.register_host(xh)
    got = [0, 0]

    def tally():
        n = _lib.lib().fdc_sinks_pdu_count(bank._h)
        if n > 0:
            arr = (_lib.fdc_pdu * n)()
            _lib.lib().fdc_sinks_pdus(bank._h, arr, n)
            got[0] += int(np.frombuffer(arr, dtype=np.dtype(_lib.fdc_pdu))["nsamples"].sum())
            got[1] += n

    def lap():
        for k in range(calls_per_lap):
            pipe.work_sinks_raw(xh.ctypes.data + 8 * k * per_call * H, per_call, none, bank)
            tally()
    try:
        lap()
        while pipe.flush_sinks(bank) > 0:
            pass
        got[:] = [0, 0]
        laps = max(2, (8 + calls_per_lap - 1) // calls_per_lap)
        t0 = time.perf_counter()
        for _ in range(laps):
            lap()
        while pipe.flush_sinks(bank) > 0:                    # what is still inside belongs to the region
            tally()
        dt = time.perf_counter() - t0
        ncalls = laps * calls_per_lap
        return {"value": round(ncalls * per_call * H / dt / 1e6, 3), "unit": "Msamples/s", "blocks_per_call": per_call, "calls": ncalls,
                "ms_per_call": round(dt / ncalls * 1e3, 3), "pdus_per_call": round(got[1] / ncalls, 1),
                "extracted_samples_per_call": round(got[0] / ncalls, 1),
                "pdu_latency_calls": pipe.sinks_latency(bank), "sink_engine": "device" if bank.engine() == 1 else "host",
                "entry": "fdc_pipeline_work_sinks, %s (H2D of the items from a buffer pinned with fdc_host_register + forward transform + "
                         "sinks + PDU payloads to host memory per call)" %
                         ("PIPELINED: look-ahead bank, front end of call n beside the sinks of call n - 1" if pipelined else
                          "serial: front end, then the sinks, inside the call")}
    finally:
        G.unregister_host(xh)
        bank.close()
        pipe.close()


def main():
    a = parse()
    dry = os.environ.get("FDC_BENCH_DRYRUN") == "1"      # launcher rehearsal on a CPU box: ranks, barrier, MAX — no GPU work
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(a))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit("--gpus %d but %d rank(s) are running" % (a.gpus, world))

    # every rank on the CPUs of its GPU's NUMA node, before anything touches the GPU (the host-buffer legs copy from this process's memory)
    placed = None
    if world > 1 and not dry and os.environ.get("FDC_BENCH_REHEARSE") != "1":
        placed = gpu_numa_cpus(local)
        if placed:
            try:
                os.sched_setaffinity(0, placed[1])
            except OSError:                       # best effort: a cpuset that refuses the mask leaves the rank where it is
                placed = None
    import torch                      # first: my library then binds to the same libamdhip64 torch loaded
    import numpy as np
    # rehearsal on a one-GPU box (FDC_BENCH_REHEARSE=1): all ranks share cuda:0 and talk over gloo; the driver's real
    # multi-GPU run is one rank per GPU over RCCL
    rehearse = os.environ.get("FDC_BENCH_REHEARSE") == "1"
    dist = None
    if dry:
        import torch.distributed as dist
        if world > 1:
            dist.init_process_group("gloo")
        t0 = time.perf_counter()
        time.sleep(0.01 * (rank + 1))
        dt = time.perf_counter() - t0
        per_rank_ms, seen, backend = [round(dt * 1e3, 4)], 1, None
        if world > 1:
            dist.barrier()
            mine = torch.tensor([dt], dtype=torch.float64)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            per_rank_ms = [round(float(x.item()) * 1e3, 4) for x in every]
            t = mine.clone()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            seen, backend = dist.get_world_size(), dist.get_backend()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"metric": METRIC, "value": 0.0, "unit": "Msamples/s", "n_gpus": world, "steps": a.steps,
                              "warmup": a.warmup, "ms_per_step": round(dt * 1e3, 4), "higher_is_better": True, "scaling": "weak",
                              "vs_baseline": None, "dtype": "f32", "data": "dry-run: launcher rehearsal, no GPU work",
                              "config": {"workload": "none (FDC_BENCH_DRYRUN=1)", "ranks_seen": seen, "backend": backend,
                                         "per_rank_ms": per_rank_ms}}))
        return
    import gr_fdc_amd as G
    if a.force_path:
        G.defaults[{"no-block": "FDC_NO_BLOCK", "no-poly": "FDC_NO_POLY", "generic": "FDC_FORCE_GENERIC",
                    "full-spectrum": "FDC_FULL_SPECTRUM", "wide-uniform": "FDC_WIDE_UNIFORM", "no-fused": "FDC_NO_FUSED"}[a.force_path]] = "1"
    if a.block_hints is not None:
        G.defaults["FDC_BLOCK_HINTS"] = str(a.block_hints)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    ndev = torch.cuda.device_count()
    if rehearse:
        local = local % max(1, ndev)
    elif local >= ndev:
        raise SystemExit("rank %d has no GPU of its own (%d visible): refusing to share one" % (rank, ndev))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    N, R, C, nb = a.blocklen, a.relinvovl, a.channels, a.blocks
    if a.lookahead:
        assert a.config in (3, 5) and not a.sync_sinks, "--lookahead applies to the sink configurations' two-deep form"
        wgs = torch.cuda.get_device_properties(dev).multi_processor_count - a.reserve_cus
        nb = a.blocks = max(wgs, nb // wgs * wgs)      # whole rounds of the persistent kernel's workgroups (1024 -> 992 on 248)
    H = N - N // R
    first_block, _n = G.span_for_rank(world * nb, rank, world)   # contiguous span per rank (§8e); weak scaling
    sinks, segments = None, None
    extracted = [0, 0, 0]                              # cfg3 / cfg5: samples, PDUs and batches handed out in the timed region
    if a.config in (1, 2, 4):
        # channel plan through the reference's own parameter derivation (py:322-345): tiles the spectrum
        if a.config == 1:    # examples/FDC_example.grc: [[0.12,0.05],[0.22,0.1],[-0.14,0.12],[0,0.081]]
            params = [G.get_opt_channelparams(N, R, (u + 0.5) % 1.0, bw) for (u, bw) in
                      ((0.12, 0.05), (0.22, 0.1), (-0.14, 0.12), (0.0, 0.081))]
        elif a.sparse:
            wid = tuple(int(v) for v in a.sparse_widths.split(","))
            params = [G.get_opt_channelparams(N, R, ((c + 0.37) / a.sparse) % 1.0, 0.8 * wid[c % len(wid)] / N) for c in range(a.sparse)]
        elif a.gapless:     # C channels of 1/C of the band each: the derivation doubles the slices (l = 2 N / C), neighbours overlap by half
            C = a.gapless
            params = [G.get_opt_channelparams(N, R, ((c + (0.0 if a.centred else 0.5)) / C) % 1.0, 1.0 / C) for c in range(C)]
        elif a.two_widths:
            assert N == 65536, "--two-widths is defined at N = 65536"
            params = [(256 * c, 256, 256 - 256 // R, 0.88, 1.0) for c in range(128)] + [(512 * c, 512, 512 - 512 // R, 0.88, 1.0) for c in range(64, 128)]
            C = len(params)
        elif a.width:
            C = N // a.width
            if a.centred:     # centres on k/C: half a channel off the grid, channel 0 wrapped and clamped onto it by the reference's derivation
                params = [G.get_opt_channelparams(N, R, (c / C) % 1.0, 0.8 / C) for c in range(C)]
            else:
                params = [G.get_opt_channelparams(N, R, ((c + 0.5) / C) % 1.0, 0.8 / C) for c in range(C)]
                assert all(p_[1] == a.width and p_[0] == a.width * c for c, p_ in enumerate(params)), "not a bank on the %d-bin grid" % a.width
        else:
            bws = (0.8, 0.4, 0.8, 1.6) if a.mixed else (0.8,)
            params = [G.get_opt_channelparams(N, R, ((c + (0.0 if a.centred else 0.5)) / C - 0.5 + 0.5) % 1.0, bws[c % len(bws)] / C) for c in range(C)]
        if a.offset:
            params = [(f + a.offset, l, lo, p, s) for (f, l, lo, p, s) in params[:-1]]
        if a.extra:
            for k in range(a.extra):
                l = (512, 128, 1024)[k % 3]
                params.append((int((k + 0.37) / a.extra * (N - 2048)) | 1, l, l - l // R, 0.7, 0.9))
        plan = [(f, l, p, s) for (f, l, _lo, p, s) in params]
        sum_lout = sum(lo for (_f, _l, lo, _p, _s) in params)
        pipe = G.Pipeline(N, R, plan, windowtype=1, max_blocks=nb, device_id=local, chunk_blocks=a.chunk)
        x = synth_input(torch, dev, N, R, C, nb, first_block, 2025 + rank, carriers=not a.noise_input)
        # further rings: the same multicarrier signal with the samples rolled (distinct addresses and distinct data; the
        # generator itself takes seconds per ring)
        rings = [x] + [torch.roll(x, 7919 * (i + 1)) for i in range(max(1, a.input_rings) - 1)]
        out = torch.empty(pipe.output_samples(nb), dtype=torch.complex64, device=dev)
        wl = "%s: %d-pt FFT, 1/%d overlap-save, %d fixed channels (l=%d, lout=%d), %d blocks/step/GPU" % (
            "configs[0] (example flowgraph plan)" if a.config == 1 else
            "configs[1]" if (N, R, C) == (65536, 2, 256) else "configs[3] per-GPU shape" if (N, R, C) == (262144, 2, 1024)
            else "non-default shape", N, R, len(plan), params[0][1], params[0][2], nb) + \
            (", widths l = %s as the reference's derivation gives them" % [p_[1] for p_ in params] if a.config == 1 else "") + (", offset %d bins" % a.offset if a.offset else "") + \
            (", %d input rings in rotation (cache-cold input)" % len(rings) if len(rings) > 1 else ", ONE input ring (stays in the memory-side cache)") + \
            (", MIXED widths l = %s" % sorted(set(p_[1] for p_ in params)) if a.mixed else "") + \
            (", TWO BANKS: 128 channels of 256 bins + 64 channels of 512 bins" if a.two_widths else "") + \
            (", plus %d channels of other widths at odd bins (split plan)" % a.extra if a.extra else "") + \
            (", SPARSE plan: %d channels, l = %s, %d of %d bins read" % (len(plan), sorted(set(p_[1] for p_ in params)), sum(p_[1] for p_ in params), N)
             if a.sparse else "")
    else:
        # the stateful sinks run on a spectrum in device memory: forward transform into the bank's buffer, then the bank
        plan, params, sum_lout = [], [], 0
        pipe = G.Pipeline(N, R, [], windowtype=1, max_blocks=nb, device_id=local, chunk_blocks=a.chunk, keep_spectrum=True)
        if a.config == 3:
            pac = [(((c + 0.5) / C) % 1.0, 0.8 / C, c) for c in range(C)]

            def make_bank(max_blocks, lookahead, device_payload):
                return G.Sinks(N, R, pac=pac, pac_thresh=6.0, pac_maxblocks=128, pac_delay=1, max_blocks=max_blocks, device_id=local,
                               host_decisions=a.sink_engine == "host", device_payload=device_payload, lookahead=lookahead)
            sinks = make_bank(nb, a.lookahead, a.payload == "device")
            carriers = [((c + 0.5) / C - 0.5, 1.0 / C) for c in range(C)]
            wl = "configs[2]: %d-pt FFT, 1/%d overlap-save, %d PowerActivationChannel sinks (6 dB, maxblocks 128), bursty " \
                 "carriers (8-64 blocks, 50 %% duty, 30 dB), %d blocks/step" % (N, R, C, nb)
        else:
            segments = [((0.05 + 0.5) % 1.0, (0.45 + 0.5) % 1.0), ((-0.45 + 0.5) % 1.0, (-0.05 + 0.5) % 1.0)]

            def make_bank(max_blocks, lookahead, device_payload):
                return G.Sinks(N, R, segments=segments, det_thresh=10.0, det_maxblocks=128, minchandist=0.005, det_delay=1,
                               puffer=0.2, max_blocks=max_blocks, device_id=local, host_decisions=a.sink_engine == "host",
                               device_payload=device_payload, lookahead=lookahead)
            sinks = make_bank(nb, a.lookahead, a.payload == "device")
            rng = np.random.default_rng(2028)
            carriers, used = [], []
            while len(carriers) < 24:                  # 24 carriers of width 0.002-0.03 at non-overlapping centres inside the segments
                wd = float(rng.uniform(0.002, 0.03))
                lo, hi = ((0.05, 0.45), (-0.45, -0.05))[int(rng.integers(0, 2))]
                fc = float(rng.uniform(lo + wd, hi - wd))
                if all(abs(fc - u) > (wd + v) * 0.75 + 0.006 for (u, v) in used):
                    used.append((fc, wd)); carriers.append((fc, wd))
            wl = "configs[4]: %d-pt FFT, 1/%d overlap-save, activity_detection_channelizer_vcm, segments [0.05,0.45] and " \
                 "[-0.45,-0.05], 10 dB, minchandist 0.005, 24 bursty carriers (widths 0.002-0.03), %d blocks/step" % (N, R, nb)
        x = synth_bursty(torch, dev, N, R, carriers, nb, 2026 if a.config == 3 else 2028)
        out = None
    b_alg = 8 * H + 8 * sum_lout                       # SURVEY.md §8d, bytes per input block (sinks: + the extracted samples)
    torch.cuda.synchronize()

    if a.check and rank == 0 and sinks is None:
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

    if sinks is None:
        ring_ptrs = [r.data_ptr() for r in rings]
        turn = [0]

        def step():
            pipe.process_device(ring_ptrs[turn[0] % len(ring_ptrs)], first_block, nb, out.data_ptr())
            turn[0] += 1
    else:
        from gr_fdc_amd import _lib
        sstream = _lib.lib().fdc_sinks_stream(sinks._h)
        count = [False]
        fused = not a.no_fused_cells

        def tally():
            n = _lib.lib().fdc_sinks_pdu_count(sinks._h)
            if n > 0:
                arr = (_lib.fdc_pdu * n)()
                _lib.lib().fdc_sinks_pdus(sinks._h, arr, n)
                extracted[0] += int(np.frombuffer(arr, dtype=np.dtype(_lib.fdc_pdu))["nsamples"].sum())
                extracted[1] += n

        if a.lookahead:
            # the bank's look-ahead form (include/fdc_amd.h, FDC_SINKS_LOOKAHEAD): every step transforms the NEXT batch on the fill stream —
            # forward transform + power cells, beside this batch's decision chains — and submits the one transformed a step ago
            fstream = sinks.fill_stream()
            assert pipe.reserve_compute_units(a.reserve_cus) == wgs
            pipe.process_device(x.data_ptr(), first_block, nb, None, d_spectrum=sinks.spectrum_ptr(), stream=fstream,
                                d_group_power=sinks.group_power_ptr() if fused else None)
            sinks.prepare(nb, ahead=False, from_groups=fused)

        def step():
            # forward transform of the batch straight into the bank's spectrum buffer (the bank's stream), then the bank:
            # power cells -> decisions -> extraction of the active (block, channel) pairs -> PDUs.  Two deep: the PDUs handed
            # out by a step are those of the batch before, whose payload copy ran beside this batch's kernels.
            # round 6 (fused): the forward kernel leaves the power of the spectrum's 16-bin groups beside the spectrum and the bank sums its cells
            # from them (fdc_sinks_prepare_from_groups) instead of reading the spectrum back (k_cell_power); --no-fused-cells: the round-5 form
            if a.lookahead:
                pipe.process_device(x.data_ptr(), first_block, nb, None, d_spectrum=sinks.spectrum_ahead_ptr(), stream=fstream,
                                    d_group_power=sinks.group_power_ahead_ptr() if fused else None)
                sinks.prepare(nb, ahead=True, from_groups=fused)
            else:
                pipe.process_device(x.data_ptr(), first_block, nb, None, d_spectrum=sinks.spectrum_ptr(), stream=sstream,
                                    d_group_power=sinks.group_power_ptr() if fused else None)
                if fused:
                    sinks.prepare(nb, ahead=False, from_groups=True)
            if a.sync_sinks:
                done = _lib.check(_lib.lib().fdc_sinks_work_device(sinks._h, nb))
            else:
                done = _lib.check(_lib.lib().fdc_sinks_submit_device(sinks._h, nb))
            if count[0] and done > 0:
                tally()
                extracted[2] += 1

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # The device leaves its idle clocks only under load: the first ~100 ms of launches run up to 7 % slower (measured: 20 timed steps behind
    # 5 / 50 / 300 warm-up steps: 0.357 / 0.345 / 0.332 ms per step).  Whatever W is, the path first runs untimed until that is over; then the W
    # warm-up steps, then the K timed ones.
    nsettle, t_s = 0, time.perf_counter()
    while (time.perf_counter() - t_s) * 1e3 < a.settle_ms:
        for _ in range(8):
            step()
        torch.cuda.synchronize()
        nsettle += 8
    for _ in range(a.warmup):
        step()
    if sinks is not None:
        _lib.check(_lib.lib().fdc_sinks_flush(sinks._h))
    fence()
    if sinks is not None:
        count[0] = True
    pipe.enable_timing(0 if a.no_kernel_timing else max(1, a.timing_stride))
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    if sinks is not None and _lib.check(_lib.lib().fdc_sinks_flush(sinks._h)) > 0:     # the batch still in flight belongs to the region
        tally()
        extracted[2] += 1
    fence()
    dt = time.perf_counter() - t0
    # per-kernel HIP-event durations, summed over every launch of the timed region (the events sit on the
    # stream the kernels are launched on); read out after the region is closed
    last = pipe.last_kernel_ms()
    pipe.enable_timing(False)
    per_rank_ms = None
    if dist is not None:
        # every rank's own time goes into the line (a straggler shows; the gather proves the backend saw `world` ranks), the job's time is the MAX
        mine = torch.tensor([dt], device="cpu" if rehearse else dev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_ms = [round(float(x.item()) / a.steps * 1e3, 4) for x in every]
        t = mine.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # The line certifies itself: blocks of the LAST timed launch's output (first, last and random ones; every 16th channel) and the
    # matching piece of the input ring it read go to the host and through the oracle (the checker; nothing of it is timed).
    verified = None
    if sinks is None and not a.no_verify:
        verified = verify_last_launch(torch, np, a, pipe, rings[(turn[0] - 1) % len(rings)], out, plan, params, N, R, nb, first_block, rank)
        if dist is not None:
            t = torch.tensor([verified["max_rel_err"]], device="cpu" if rehearse else dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            verified["max_rel_err"] = float(t.item())
            verified["ranks"] = world
    elif sinks is not None:
        verified = {"blocks": 0, "note": "sink configurations are not re-checked inside bench.py: tests/test_fullsize_gpu.py compares every PDU "
                                         "of this workload at this size with the oracle"}

    # configs[1] only: the same steps again on ONE ring (rounds 1-2 measured this: the ring then stays in the 256 MiB memory-side
    # cache from step to step) — reported beside the headline for continuity, never as `value`
    one_ring = None
    if sinks is None and len(rings) > 1 and a.config == 2 and a.one_ring_leg:
        turn[0] = 0
        ring_ptrs[:] = ring_ptrs[:1]
        small = min(nb, 1024)
        for _ in range(5):
            pipe.process_device(ring_ptrs[0], first_block, small, out.data_ptr())
        fence()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            pipe.process_device(ring_ptrs[0], first_block, small, out.data_ptr())
        fence()
        d1 = time.perf_counter() - t1
        if dist is not None:
            t = torch.tensor([d1], device="cpu" if rehearse else dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d1 = float(t.item())
        one_ring = {"blocks_per_step": small, "ms_per_step": round(d1 / a.steps * 1e3, 4),
                    "Msamples_per_s": round(world * small * H * a.steps / d1 / 1e6, 3),
                    "pipeline_frac": round((8 * H + 8 * sum_lout) * small * a.steps / d1 / 1e9 / HBM_PEAK_GBS, 4),
                    "note": "input re-read from the memory-side cache (268 MB ring, 256 MiB cache): the round-1/2 default"}
    # "also report an end-to-end number with H2D": the host-buffer entry sync_block::work() calls
    # (fdc_pipeline_work: input H2D, kernels, outputs D2H), 256 blocks per call, caller buffers pinned once with
    # fdc_host_register.  PCIe-bound; never `value`.  Every rank runs it on its own device at the same time (N ranks = N PCIe
    # links); with one rank the same call also goes through the multi-device handle over every visible device.
    # (Its sub-batches of 32 blocks take the tiled kernels, not the block kernel: the rocprofv3 average of the dominant kernel over
    # this command is still the average of the settle / warm-up / timed launches.)
    end_to_end, end_to_end_group = None, None
    if sinks is None and a.config in (1, 2) and not a.no_end_to_end and not a.no_kernel_timing:
        fence()
        e2e_blocks = 256 if N >= 65536 else max(256, (256 * 65536) // N)      # about the same bytes per call whatever the block length
        end_to_end = host_entry_leg(G, np, N, R, plan, sum_lout, [local], e2e_blocks)
        if dist is not None:
            t = torch.tensor([end_to_end["value"]], device="cpu" if rehearse else dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            end_to_end["per_rank_value_rank0"] = end_to_end["value"]
            end_to_end["value"] = round(float(t.item()), 3)
            end_to_end["ranks"] = world
        elif not rehearse:
            devs = list(range(ndev)) if ndev > 1 else [0, 0]
            end_to_end_group = host_entry_leg(G, np, N, R, plan, sum_lout, devs, e2e_blocks)
            if ndev == 1:
                end_to_end_group["note"] = "one GPU visible: two VIRTUAL members on device 0 share its one PCIe link (dispatcher exercised, no gain expected)"
    # configs[2] / configs[4] from HOST samples (round 6): the whole hier block behind ONE work()-level entry, fdc_pipeline_work_sinks on a
    # look-ahead bank = the pipelined form (include/fdc_amd.h): H2D of 256 KiB per item, forward transform, the sinks, PDUs to host memory.
    if sinks is not None and not a.no_end_to_end and not a.sync_sinks and world == 1:
        fence()
        end_to_end = [sinks_host_entry_leg(G, np, _lib, N, R, local, x, nb, make_bank, per_call, pipelined)
                      for (per_call, pipelined) in ((256, True), (min(nb, 1024), True), (256, False))]
    msps = world * nb * H * a.steps / dt / 1e6
    chunk = pipe.chunk_blocks()
    ngroups = max(1, int(last[3]))              # launch groups that carried events (every timing_stride-th of the region)
    nlaunch = (nb + chunk - 1) // chunk         # launch groups per step
    path = pipe.path()
    names = ["fused4096(FFT+cut+window+IFFTs, spectrum in LDS)", "unused", "unused2"] if path == 5 and sinks is None else \
            ["poly_stage1_generic(colFFT+window+IFFT, any width)", "poly_stage2_generic(slotFFT)", "unused"] if path == 2 and sinks is None and a.width else \
            ["block_kernel(the tilings of a split plan)", "block_fft(forward, partial spectrum)", "channels(remainder)"] if path == 4 and sinks is None else \
            ["block_kernel(colFFT+window+IFFT+slotFFT)", "unused", "unused2"] if path == 3 and sinks is None else \
            ["poly_stage1(colFFT+window+IFFT)", "poly_stage2(slotFFT)", "unused"] if path == 2 and sinks is None else \
            ["block_fft(forward, one kernel)", "unused", "channels"] if path == 1 and N == 65536 and a.force_path != "no-block" else \
            ["fft_pass_a", "fft_pass_b", "channels"]
    if sinks is not None:
        b_alg += 8.0 * extracted[0] / max(1, extracted[2] * nb)     # the data-dependent part, counted by the harness
    dom = max(range(3), key=lambda i: last[i])
    dom_avg_ms = last[dom] / ngroups            # average duration of ONE launch of the dominant kernel
    blocks_per_launch = nb / nlaunch            # units one launch processes
    achieved = b_alg * blocks_per_launch / (dom_avg_ms * 1e-3) / 1e9 if dom_avg_ms > 0 else 0.0
    # Paths of several kernels: `frac` puts ALL algorithmic bytes over ONE kernel's time (the contract's formula) and reads high; what each
    # kernel moves of its OWN traffic per second is the number to compare with a copy.  Two-launch uniform path: stage 1 reads the new
    # samples and writes G (lout * N / l samples per block), stage 2 reads G and writes the output samples.
    own = None
    if path == 2 and sinks is None and last[0] > 0 and last[1] > 0:
        g_bytes = 8.0 * sum_lout if not a.width else 8.0 * (a.width - a.width // R) * (N // a.width)
        own = {names[0]: {"bytes_per_block": 8.0 * H + g_bytes, "GB_per_s": round((8.0 * H + g_bytes) * blocks_per_launch / (last[0] / ngroups * 1e-3) / 1e9, 1)},
               names[1]: {"bytes_per_block": g_bytes + 8.0 * sum_lout, "GB_per_s": round((g_bytes + 8.0 * sum_lout) * blocks_per_launch / (last[1] / ngroups * 1e-3) / 1e9, 1)},
               "note": "own traffic of each kernel (G out and back counts here, not in b_alg): compare these with the copy rate, pipeline_frac with the target"}
    # HBM bytes per launch of the dominant kernel from rocprofv3 PMC passes of this same command
    # (profiles/pmc_run.sh -> profiles/pmc_traffic.json; FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM)
    traffic, traffic_source = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            pt = json.load(fh)
        ent = pt.get("cfg%d/%s" % (a.config, names[dom]))
        if ent and ent.get("blocks_per_launch") == blocks_per_launch and ent.get("blocklen") == N and not (a.offset or a.mixed or a.sparse or a.extra or a.width or a.gapless or a.centred or a.two_widths or R != 2):
            traffic = ent["hbm_bytes_per_launch"]
            traffic_source = "profiles/pmc_traffic.json: rocprofv3 --pmc passes of this command on another run (profiles/pmc_run.sh), " \
                             "not counters of this process"
    except (OSError, ValueError):
        pass
    pipe_gbs = b_alg * nb * a.steps / dt / 1e9
    multi_kernel = sinks is not None or sum(1 for v in last[:3] if v > 0) > 1 or nlaunch > 1
    res = {
        "metric": METRIC,
        "value": round(msps, 3), "unit": "Msamples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": wl, "baseline_config": a.config,
                   "blocklen": N, "relinvovl": R, "channels": C, "blocks_per_step_per_gpu": nb,
                   "input_rings": len(rings) if sinks is None else 1,
                   "settle_ms": a.settle_ms, "settle_steps": nsettle,
                   # what ran untimed in front of the K timed steps: the running-in (the device leaves its idle clocks only under load)
                   # plus the W warm-up steps the command line names
                   "effective_warmup_steps": nsettle + a.warmup,
                   # revision of the headline workload: 1 = rounds 1-2 (1024 blocks per step on ONE 268 MB ring, input re-read from the
                   # memory-side cache), 2 = round 3 on (2048 blocks per step, three 537 MB rings in rotation: cache-cold input, 150 ms
                   # running-in).  Lines of different revisions are not comparable.
                   "workload_rev": 2,
                   "chunk_blocks": chunk, "kernel_path": path, "kernel_plan": pipe.describe(), "parallelism": "block-span sharding x%d, no collective" % world,
                   # N > 1: what the process group reports (not what the command line asked for) and every rank's own step time
                   "ranks_seen": dist.get_world_size() if dist is not None else 1,
                   "backend": dist.get_backend() if dist is not None else None,
                   "per_rank_ms": per_rank_ms,
                   "rank0_numa_node": placed[0] if placed else None},
        # frac: the contract's definition (all algorithmic bytes of a launch over the dominant kernel's launch time);
        # pipeline_frac: SURVEY.md §8d's headline, algorithmic bytes over the WHOLE step.  With the one-kernel path (3) the
        # dominant kernel IS the step, and the two coincide up to the launch gaps.
        # Lines of SEVERAL kernels (two launches, spectrum path, split plans, the sinks): all algorithmic bytes over ONE kernel's time reads high
        # (configs[3]: 0.61 for a step at 0.31) — `achieved` / `frac` are the step's there (= pipeline_*), the contract's one-kernel formula stays
        # beside them as dominant_kernel_*, each kernel's own rate in own_traffic_rate.
        "roofline": {"bound": "hbm", "kernel": names[dom],
                     "achieved": round(pipe_gbs if multi_kernel else achieved, 2), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round((pipe_gbs if multi_kernel else achieved) / HBM_PEAK_GBS, 4),
                     "frac_is": "whole step (several kernels per step)" if multi_kernel else "dominant kernel = the step (one launch per step)",
                     "dominant_kernel_achieved": round(achieved, 2), "dominant_kernel_frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": traffic, "traffic_source": traffic_source,
                     "kernel_ms_per_step": {n: round(v / ngroups * nlaunch, 4) for n, v in zip(names, last) if not n.startswith("unused")},
                     "kernel_avg_launch_ms": round(dom_avg_ms, 5), "blocks_per_launch": blocks_per_launch,
                     "launches_per_step": nlaunch, "timed_launches": ngroups, "timing_stride": max(1, a.timing_stride),
                     "alg_bytes_per_block": b_alg, "own_traffic_rate": own,
                     "pipeline_achieved": round(pipe_gbs, 2),
                     "pipeline_frac": round(pipe_gbs / HBM_PEAK_GBS, 4),
                     # SURVEY.md §8d also asks for the fraction of the achievable float4-copy rate (6.3 TB/s per the guide)
                     "one_ring": one_ring,
                     "frac_of_achievable_6300": round(achieved / 6300.0, 4),
                     "pipeline_frac_of_achievable_6300": round(pipe_gbs / 6300.0, 4)},
    }
    if verified is not None:
        res["verified"] = verified
    if end_to_end is not None:
        res["end_to_end_h2d"] = end_to_end
    if end_to_end_group is not None:
        res["end_to_end_h2d_group"] = end_to_end_group
    if sinks is not None:
        res["config"]["pdus_per_step"] = round(extracted[1] / max(1, extracted[2]), 1)
        res["config"]["extracted_samples_per_step"] = round(extracted[0] / max(1, extracted[2]), 1)
        res["config"]["batches_with_pdus_read"] = extracted[2]
        res["config"]["sink_engine"] = "device" if sinks.engine() == 1 else "host"
        res["config"]["payload"] = a.payload if sinks.engine() == 1 else "host"
        res["config"]["submission"] = "synchronous" if a.sync_sinks else "two deep (fdc_sinks_submit_device)"
        res["config"]["power_cells"] = "from the forward kernel's 16-bin group sums (fdc_sinks_prepare_from_groups)" if fused else "pass over the spectrum (k_cell_power)"
        if a.lookahead:
            res["config"]["submission"] += ", look-ahead: forward transform + power cells of batch n + 1 on the fill stream beside batch n's decisions"
            res["config"]["lookahead"] = {"reserved_compute_units": a.reserve_cus, "block_kernel_workgroups": wgs}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        if sinks is None:
            _model, _nproc, share = host_info()
            res["cpu_baseline"] = cpu_baseline(N, R, plan, a.cpu_blocks or max(share * 4, 64), a.cpu_budget)
        else:
            res["cpu_baseline"] = cpu_baseline_sinks(a, N, R, C, x, nb, segments)
    if rank == 0:
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()
    if verified is not None and verified.get("blocks", 0) > 0 and not verified["max_rel_err"] <= 1e-5:
        raise SystemExit("bench.py: the last timed launch's output differs from the oracle: max rel err %.3g" % verified["max_rel_err"])


if __name__ == "__main__":
    main()
