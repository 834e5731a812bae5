#!/usr/bin/env python3
"""The whole hier block — front end, one throughput channel, an activity-controlled channel (PowerActivationChannel) and a detection segment
(SegmentDetection) — behind ONE work()-level entry, pipelined (round 6): every work() call copies and transforms its items beside the sink blocks
of the call before; PDUs come out two calls later; flush() at the end of the stream.  Same constructor as FDC.FrequencyDomainChannelizer
(python/FrequencyDomainChannelizer.py:46-60) plus `pipelined=True`.

  python examples/hier_pipelined.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gr_fdc_amd as G  # noqa: E402

N, R, items = 65536, 2, 64                       # block length, relative inverse overlap, items per work() call
H = N - N // R
fdc = G.FrequencyDomainChannelizer(
    8, 1, N, R,
    [[0.10, 0.004]],                             # throughput channels (centre, bandwidth; 'normalized': -0.5 .. 0.5 of fs)
    [[-0.20, 0.004]], 6.0,                       # activity-controlled channels, their threshold in dB
    1.0, 0.0, 'normalized', G.WINDOWTYPES.HANN,
    True, False, "", False,                      # message output, file output, path, threaded
    [[0.25, 0.40]], 10.0, 0.005, 1, 0.2, 0,      # detection segments, threshold, minchandist, deactivation delay, flank puffer, verbose
    1, 16, 16, False,                            # PowerActivationChannel delay, maxblocks of the two sink kinds, debug port
    max_blocks=items, pipelined=True)
print("PDUs of a call's items come out %d calls later" % fdc.pipeline.sinks_latency(fdc.sinks))

rng = np.random.default_rng(2)
ncalls = 8
n = np.arange(ncalls * items * H)
x = 0.01 * (rng.standard_normal(n.size) + 1j * rng.standard_normal(n.size))
for fc, first, last in ((-0.20, 40, 200), (0.31, 120, 330), (0.10, 0, ncalls * items)):      # bursts in blocks
    env = np.zeros(n.size)
    env[first * H:last * H] = 1.0
    x += env * np.exp(2j * np.pi * fc * n)
x = x.astype(np.complex64)

total = 0
for k in range(ncalls):
    ports = fdc.work(x[k * items * H:(k + 1) * items * H])          # stream outputs of THIS call's items
    for d, samples in fdc.messages:                                  # PDUs of the items of call k - 2
        print("call %d: %-44s blocks %4d..%4d  %7d samples" % (k, d["ID"][20:], d["blockstart"], d["blockend"], samples.size))
    total += len(fdc.messages)
for d, samples in fdc.flush():                                       # what the sink blocks still hold
    print("flush : %-44s blocks %4d..%4d  %7d samples" % (d["ID"][20:], d["blockstart"], d["blockend"], samples.size))
    total += 1
print("%d PDUs; throughput channel: %d samples per call" % (total, ports[0].size))
