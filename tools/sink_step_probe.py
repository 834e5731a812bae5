#!/usr/bin/env python3
"""GPU box helper: where a configs[2] bench step spends its time on the Python side (C calls timed one by one)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--config", "3"]
import numpy as np, torch
import bench as B
import gr_fdc_amd as G
from gr_fdc_amd import _lib
a = B.parse()
N, R, C, nb = a.blocklen, a.relinvovl, a.channels, a.blocks
dev = torch.device("cuda:0")
pipe = G.Pipeline(N, R, [], windowtype=1, max_blocks=nb, keep_spectrum=True)
pac = [((c + 0.5) / C, 0.8 / C, c) for c in range(C)]
sinks = G.Sinks(N, R, pac=pac, pac_thresh=6.0, pac_maxblocks=128, pac_delay=1, max_blocks=nb)
carriers = [(((c + 0.5) / C) - 0.5, 0.5 / C) for c in range(C)]
x = B.synth_bursty(torch, dev, N, R, carriers, nb, 2026)
sstream = _lib.lib().fdc_sinks_stream(sinks._h)
T = np.zeros(4)
for it in range(8):
    t0 = time.perf_counter()
    pipe.process_device(x.data_ptr(), 0, nb, None, d_spectrum=sinks.spectrum_ptr(), stream=sstream)
    t1 = time.perf_counter()
    _lib.check(_lib.lib().fdc_sinks_work_device(sinks._h, nb))
    t2 = time.perf_counter()
    n = _lib.lib().fdc_sinks_pdu_count(sinks._h)
    arr = (_lib.fdc_pdu * n)()
    t3 = time.perf_counter()
    _lib.lib().fdc_sinks_pdus(sinks._h, arr, n)
    tot = int(np.frombuffer(arr, dtype=np.dtype(_lib.fdc_pdu))["nsamples"].sum())
    t4 = time.perf_counter()
    if it >= 3:
        T += np.array([t1 - t0, t2 - t1, t3 - t2, t4 - t3])
print("ms per step: process_device %.3f  sinks_work_device %.3f  array alloc %.3f  pdus copy+sum %.3f  (n=%d, %d samples)" % (*(T / 5 * 1e3), n, tot))
