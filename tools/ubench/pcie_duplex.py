#!/usr/bin/env python3
"""PCIe on this box: pinned H2D alone, D2H alone, and both at once on two streams (is the link used full duplex by the
runtime's copy engines?).  96 MiB per copy, 10 copies each."""
import time
import torch

n = 96 << 20
d_in, d_out = (torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(2))
h_in, h_out = (torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(2))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(up, down, reps=10):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        if up:
            with torch.cuda.stream(s1):
                d_in.copy_(h_in, non_blocking=True)
        if down:
            with torch.cuda.stream(s2):
                h_out.copy_(d_out, non_blocking=True)
    torch.cuda.synchronize()
    return reps * n / (time.perf_counter() - t) / 1e9


for _ in range(2):
    run(True, True, 2)
print("H2D alone %.1f GB/s" % run(True, False))
print("D2H alone %.1f GB/s" % run(False, True))
print("both at once: %.1f GB/s each way" % run(True, True))
