#!/usr/bin/env python3
"""Documentation measurements beside bench.py (GPU box): PCIe-inclusive host path for cfg2, and the stateful
configurations cfg3 (256 PowerActivationChannel sinks) and cfg5 (activity_detection_channelizer_vcm) end to end
through the host-buffer entry points (spectrum stays on the device)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gr_fdc_amd as G  # noqa: E402

N, R, C = 65536, 2, 256
H = N - N // R
nb = 256
rng = np.random.default_rng(2026)
params = [G.get_opt_channelparams(N, R, ((c + 0.5) / C) % 1.0, 0.8 / C) for c in range(C)]
plan = [(f, l, p, s) for (f, l, _lo, p, s) in params]


def bursty(nblocks, ncar, seed):
    g = np.random.default_rng(seed)
    n = nblocks * H
    x = 0.003 * (g.standard_normal(n) + 1j * g.standard_normal(n))
    t = np.arange(n)
    for k in range(ncar):
        fc = (k + 0.5) / ncar - 0.5
        env = np.repeat(g.integers(0, 2, nblocks // 8 + 1), 8 * H)[:n].astype(np.float64)
        x += env * 0.05 * np.exp(2j * np.pi * ((fc * t) % 1.0))
    return x.astype(np.complex64)


def timeit(fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


x = (rng.standard_normal(nb * H) + 1j * rng.standard_normal(nb * H)).astype(np.complex64)
p = G.Pipeline(N, R, plan, windowtype=1, max_blocks=nb)
dt = timeit(lambda: p.work(x))
print("cfg2 host path (fdc_pipeline_work, pageable host buffers, fresh output arrays per call, H2D+D2H included): "
      "%.1f Msamples/s in" % (nb * H / dt / 1e6))
los = p.lout
pool = np.zeros(nb * sum(los), np.complex64)


def views(n):
    out, off = [], 0
    for lo in los:
        out.append(pool[off:off + n * lo]); off += n * lo
    return out


import ctypes as ct


def raw(n):
    ptrs = (ct.c_void_p * len(los))(*[o.ctypes.data for o in views(n)])
    return lambda: p.work_raw(x.ctypes.data, n, ptrs)


for n in (nb, 64, 8, 1):
    dt = timeit(raw(n), reps=10)
    print("cfg2 host path, pageable buffers reused, %4d blocks/call: %.1f Msamples/s in (%.3f ms/call)"
          % (n, n * H / dt / 1e6, dt * 1e3))
G.register_host(x); G.register_host(pool)
for n in (nb, 64, 8, 1):
    dt = timeit(raw(n), reps=10)
    print("cfg2 host path, buffers pinned with fdc_host_register, %4d blocks/call: %.1f Msamples/s in (%.3f ms/call)"
          % (n, n * H / dt / 1e6, dt * 1e3))
G.unregister_host(x); G.unregister_host(pool)

xb = bursty(nb, 32, 1)
pac = [(((c + 0.5) / C) % 1.0, 0.8 / C, c) for c in range(C)]
sinks = G.Sinks(N, R, pac=pac, pac_thresh=6.0, pac_maxblocks=128, pac_delay=1, max_blocks=nb)
p3 = G.Pipeline(N, R, [], windowtype=1, max_blocks=nb, keep_spectrum=True)
npdu = [0]


def run3():
    p3.work(xb, sinks=sinks)
    npdu[0] = len(sinks._collect())


dt = timeit(run3)
print("cfg3 (256 PowerActivationChannel sinks, bursty input, %d PDUs/batch): %.1f Msamples/s in" % (npdu[0], nb * H / dt / 1e6))
dt = timeit(lambda: p3.work(xb, sinks=sinks))
print("cfg3, C entry only (PDUs left in the handle, no Python objects): %.1f Msamples/s in (%.2f ms per %d blocks)" % (nb * H / dt / 1e6, dt * 1e3, nb))
xp = xb.copy()                       # its own buffer: the pageable measurements keep using xb
G.register_host(xp)
dt = timeit(lambda: p3.work(xp, sinks=sinks))
print("cfg3, C entry only, input pinned: %.1f Msamples/s in (%.2f ms per %d blocks)" % (nb * H / dt / 1e6, dt * 1e3, nb))
G.unregister_host(xp)

det = G.Sinks(N, R, segments=[(0.05, 0.45), (0.55, 0.95)], det_thresh=10.0, det_maxblocks=128, minchandist=0.005,
              det_delay=1, puffer=0.2, max_blocks=nb)
p5 = G.Pipeline(N, R, [], windowtype=1, max_blocks=nb, keep_spectrum=True)


def run5():
    p5.work(xb, sinks=det)
    npdu[0] = len(det._collect())


dt = timeit(run5)
print("cfg5 (vcm, 2 segments, dec=%d, %d PDUs/batch): %.1f Msamples/s in" % (det.segment_params(0)["dec"], npdu[0], nb * H / dt / 1e6))
