#!/usr/bin/env python3
"""GPU box helper: when does every workgroup of the headline launch start and end?  Needs the -DFDC_BLK_WGTIMES build:
  tools/build_variant.sh wgt -DFDC_BLK_WGTIMES && FDC_AMD_LIB=$PWD/gr-fdc_amd/libfdc_amd_wgt.so python tools/wg_times.py
The library prints its debug buffer on stderr at synchronize (rows of 32 values, the last of each row not shown); this script launches one step of
2048 blocks a few times, reads the rows back and prints the spread of the workgroups' run times and end times (100 MHz clock: 10 ns units)."""
import os
import re
import subprocess
import sys

if len(sys.argv) > 1 and sys.argv[1] == "child":
    os.environ["FDC_DEBUG_ENV"] = "1"
    os.environ["FDC_BLOCK_DEBUG"] = "1"
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    import torch
    import gr_fdc_amd as G
    N, R, C, nb = 65536, 2, 256, 2048
    plan = [(256 * c, 256, 0.88, 1.0) for c in range(C)]
    pipe = G.Pipeline(N, R, plan, windowtype=1, max_blocks=nb, chunk_blocks=nb)
    x = torch.randn(N // R + nb * (N - N // R), 2, device="cuda")
    out = torch.empty(pipe.output_samples(nb), dtype=torch.complex64, device="cuda")
    for _ in range(5):
        pipe.process_device(x.data_ptr(), 0, nb, out.data_ptr())
    torch.cuda.synchronize()
    pipe.synchronize()
    sys.exit(0)

r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], capture_output=True, text=True)
vals = {}
for line in r.stderr.splitlines():
    m = re.match(r"\[fdc block\] round (\d+) wave (\d+) t0=(\d+) :(.*)", line)
    if not m:
        continue
    k, w, t0 = int(m.group(1)), int(m.group(2)), int(m.group(3))
    row = (w * 4 + k) * 32
    vals[row] = t0
    for i, v in enumerate(m.group(4).split()):
        vals[row + 1 + i] = t0 + int(v)
starts = [vals[i] for i in range(256) if i in vals and (256 + i) in vals]
ends = [vals[256 + i] for i in range(256) if i in vals and (256 + i) in vals]
if not starts:
    print(r.stderr[-2000:])
    sys.exit("no stamps: is FDC_AMD_LIB the -DFDC_BLK_WGTIMES build?")
dur = [e - s for s, e in zip(starts, ends)]
t_first, t_last = min(starts), max(ends)
print("workgroups seen: %d   launch: %.1f us" % (len(dur), (t_last - t_first) / 100.0))
print("start spread: %.1f us   end spread: %.1f us (first %.1f, last %.1f after the first start)" % (
    (max(starts) - t_first) / 100.0, (t_last - min(ends)) / 100.0, (min(ends) - t_first) / 100.0, (t_last - t_first) / 100.0))
d = sorted(dur)
print("run time per workgroup, us: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f  mean %.1f" % (
    d[0] / 100.0, d[len(d) // 10] / 100.0, d[len(d) // 2] / 100.0, d[9 * len(d) // 10] / 100.0, d[-1] / 100.0, sum(d) / len(d) / 100.0))
by_xcd = {}
for i in range(256):
    if i in vals and (256 + i) in vals:
        by_xcd.setdefault(i & 7, []).append(vals[256 + i] - vals[i])
print("mean run time by XCD (workgroup mod 8), us:", " ".join("%.1f" % (sum(v) / len(v) / 100.0) for _x, v in sorted(by_xcd.items())))
