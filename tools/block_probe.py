#!/usr/bin/env python3
"""GPU box helper: one launch of the one-block-per-CU kernel with FDC_BLOCK_DEBUG=1; the library prints the
cycle stamps of workgroup 0 (per wave: end of each stage-1 pass, stage-2 phases) on stderr at synchronize.
The stamps are compiled in only with -DFDC_BLK_STAMPS (they cost registers the kernel does not have):
  tools/build_variant.sh stamps -DFDC_BLK_STAMPS && FDC_AMD_LIB=gr-fdc_amd/libfdc_amd_stamps.so python tools/block_probe.py"""
import os
import sys
os.environ["FDC_DEBUG_ENV"] = "1"
os.environ["FDC_BLOCK_DEBUG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import gr_fdc_amd as G  # noqa: E402
N, R, C, nb = 65536, 2, 256, 1024
plan = [(256 * c, 256, 0.88, 1.0) for c in range(C)]
pipe = G.Pipeline(N, R, plan, windowtype=1, max_blocks=nb)
x = torch.randn(N // R + nb * (N - N // R), 2, device="cuda")
out = torch.empty(pipe.output_samples(nb), dtype=torch.complex64, device="cuda")
for _ in range(3):
    pipe.process_device(x.data_ptr(), 0, nb, out.data_ptr())
torch.cuda.synchronize()
pipe.synchronize()
