#!/usr/bin/env python3
"""Which engine carries a pinned device-to-host copy on this box: run under
   rocprofv3 --kernel-trace --memory-copy-trace ... -- python3 tools/ubench/d2h_engine.py
and look for __amd_rocclr_copyBuffer in the kernel trace (shader copy) or an entry in the copy trace (SDMA).  Prints the rate."""
import time
import torch

n = 96 << 20
d = torch.empty(n, dtype=torch.uint8, device="cuda")
h = torch.empty(n, dtype=torch.uint8).pin_memory()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        h.copy_(d, non_blocking=True)
    s.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        h.copy_(d, non_blocking=True)
    s.synchronize()
    dt = time.perf_counter() - t
print("D2H %.1f GB/s" % (10 * n / dt / 1e9))
