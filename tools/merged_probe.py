import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import gr_fdc_amd as G
N, R, C, nb = 65536, 2, 256, 1024
H = N - N // R
params = [G.get_opt_channelparams(N, R, ((c + 0.5) / C) % 1.0, 0.8 / C) for c in range(C)]
plan = [(f, l, p, s) for (f, l, _lo, p, s) in params]
pipe = G.Pipeline(N, R, plan, windowtype=1, max_blocks=nb)
x = torch.randn(N // R + nb * H, dtype=torch.complex64, device="cuda")
out = torch.empty(pipe.output_samples(nb), dtype=torch.complex64, device="cuda")
for i in range(3):
    pipe.process_device(x.data_ptr(), 0, nb, out.data_ptr())
pipe.synchronize()
