import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gr_fdc_amd as G
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "measure_extra.py")).read().split("x = (rng.standard_normal")[0])
xb = bursty(nb, 32, 1)
det = G.Sinks(N, R, segments=[(0.05, 0.45), (0.55, 0.95)], det_thresh=10.0, det_maxblocks=128, minchandist=0.005,
              det_delay=1, puffer=0.2, max_blocks=nb)
p5 = G.Pipeline(N, R, [], windowtype=1, max_blocks=nb, keep_spectrum=True)
for it in range(4):
    t0 = time.perf_counter(); p5.work(xb, sinks=det); t1 = time.perf_counter(); pd = det._collect(); t2 = time.perf_counter()
    print("work %.2f ms  collect %.2f ms  pdus %d  samples %d" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, len(pd), sum(d.size for _, d in pd)))
