#!/usr/bin/env python3
"""Quick start: 256 channels out of a 65536-point overlap-save channelizer on one MI355X, through the Python mirror of the
reference's face (gr-fdc_amd/channelizer.py).  Needs libfdc_amd.so (python -c "import __graft_entry__ as g; g.build()").

  python examples/channelize.py
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gr_fdc_amd as G  # noqa: E402

N, R, C, nblocks = 65536, 2, 256, 256
H = N - N // R                                                      # new samples per block (overlap_save)
# get_opt_channelparams (python/FrequencyDomainChannelizer.py:322-345): (f, l, lout, passbw, stopbw) per channel; its frequency
# argument counts from the lower band edge (0.5 = DC), so channel c is centred at (c + 0.5) / C - 0.5 of the sample rate
params = [G.get_opt_channelparams(N, R, ((c + 0.5) / C) % 1.0, 0.8 / C) for c in range(C)]
plan = [(f, l, pbw, sbw) for (f, l, _lout, pbw, sbw) in params]
pipe = G.Pipeline(N, R, plan, windowtype=1, max_blocks=nblocks)
print("kernel path:", pipe.path(), "(3 = one kernel per step, 2 = two launches, 1 = spectrum in memory, 0 = generic)")

rng = np.random.default_rng(1)
x = (rng.standard_normal(nblocks * H) + 1j * rng.standard_normal(nblocks * H)).astype(np.complex64)
x += np.exp(2j * np.pi * 0.1234 * np.arange(x.size)).astype(np.complex64)      # one carrier at 0.1234 fs

outs = pipe.work(x)                                                 # host buffers in, one stream per channel out
t0 = time.perf_counter()
outs = pipe.work(x)
dt = time.perf_counter() - t0
power = np.array([float(np.mean(np.abs(o) ** 2)) for o in outs])
k = int(np.argmax(power))
print("%d channels x %d samples each; strongest channel %d (centre %+.4f fs), %.1f dB over the median"
      % (len(outs), outs[0].size, k, (k + 0.5) / C - 0.5, 10 * np.log10(power[k] / np.median(power))))
print("host-buffer call (H2D + kernels + D2H): %.2f ms for %d blocks = %.2f Gsamples/s in" % (dt * 1e3, nblocks, nblocks * H / dt / 1e9))
