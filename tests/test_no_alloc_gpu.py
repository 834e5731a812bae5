"""GPU test (-m gpu): no hipMalloc / hipFree / hipHostMalloc / hipHostFree in the steady state of any work()-level entry (the debug
spectrum of work(), work_real(), work_spectrum() used to be allocated and freed per call, fdc_fft_vcc allocated four buffers and rebuilt its table per
call).  A counting shim (tests/cpp/hip_alloc_counter.c) is loaded with RTLD_GLOBAL in a FRESH process before the library, so the library's calls bind
to it; after three warm-up calls (buffers that grow with the data have grown) fifty more calls of each entry must not allocate or free anything."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes as C, os, sys
import numpy as np
shim = C.CDLL(sys.argv[1], mode=C.RTLD_GLOBAL)
sys.path.insert(0, sys.argv[2])
import gr_fdc_amd as G
from gr_fdc_amd import _lib
lib = G.lib()

def counts():
    v = (C.c_long * 4)()
    shim.fdc_test_alloc_counts(v)
    return list(v)

assert counts()[0] == 0
N, R, nb = 4096, 2, 8
H = N - N // R
rng = np.random.default_rng(3)
x = (rng.standard_normal(nb * H) + 1j * rng.standard_normal(nb * H)).astype(np.complex64)
n = np.arange(nb * H)
x += (np.exp(2j * np.pi * -0.2 * n) * (n > 2 * H) * (n < 6 * H)).astype(np.complex64)
plan = [(301, 64, 0.6, 0.85), (0, 256, 0.8, 1.0), (1024 + 7, 512, 0.8, 0.95)]
entries = {}
p = G.Pipeline(N, R, plan, max_blocks=nb, keep_spectrum=True)
assert counts()[0] > 0                                   # the shim sees the library's allocations
entries["fdc_pipeline_work + debug spectrum"] = lambda: p.work(x, want_spectrum=True)
pr = G.Pipeline(N, R, plan, max_blocks=nb, keep_spectrum=True)
xr = x.real.copy()
entries["fdc_pipeline_work_real + debug spectrum"] = lambda: pr.work_real(xr, want_spectrum=True)
ps = G.Pipeline(N, R, plan, max_blocks=nb, keep_spectrum=True)
spec_items = (rng.standard_normal(nb * N) + 1j * rng.standard_normal(nb * N)).astype(np.complex64)
entries["fdc_pipeline_work_spectrum (no sinks) + spectrum"] = lambda: ps.work_spectrum(spec_items, want_spectrum=True)
kw = dict(pac=[(0.3, 0.04, 0)], pac_thresh=6.0, pac_maxblocks=3, segments=[(0.55, 0.9)], det_thresh=10.0, det_maxblocks=3, minchandist=0.01, max_blocks=nb)
p1, b1 = G.Pipeline(N, R, plan, max_blocks=nb, keep_spectrum=True), G.Sinks(N, R, **kw)
entries["fdc_pipeline_work_sinks (serial)"] = lambda: (p1.work(x, sinks=b1), b1.pdus())
p2, b2 = G.Pipeline(N, R, plan, max_blocks=nb, keep_spectrum=True), G.Sinks(N, R, lookahead=True, **kw)
entries["fdc_pipeline_work_sinks (pipelined)"] = lambda: (p2.work(x, sinks=b2), b2.pdus())
b3 = G.Sinks(N, R, **kw)
entries["fdc_sinks_work"] = lambda: b3.work(spec_items * 1e-3)
fin = (rng.standard_normal(16 * 1024) + 1j * rng.standard_normal(16 * 1024)).astype(np.complex64)
fout = np.empty_like(fin)
entries["fdc_fft_vcc"] = lambda: _lib.check(lib.fdc_fft_vcc(0, 1024, 1, 1, fin.ctypes.data, 16, fout.ctypes.data))
os_blk = G.overlap_save(8, N, N // R)
entries["fdc_overlap_save_work"] = lambda: os_blk.work(x.view(np.uint8))
bad = []
for name, call in entries.items():
    for _ in range(3):
        call()
    before = counts()
    for _ in range(50):
        call()
    after = counts()
    print(name, [a - b for a, b in zip(after, before)])
    if after != before:
        bad.append((name, [a - b for a, b in zip(after, before)]))
while p2.flush_sinks(b2) > 0:
    pass
assert not bad, bad
print("OK")
'''


def test_no_allocation_in_the_steady_state_of_any_entry(tmp_path):
    shim = str(tmp_path / "libhipcount.so")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", os.path.join(ROOT, "tests", "cpp", "hip_alloc_counter.c"), "-o", shim,
                           "-ldl", "-L/opt/rocm/lib", "-Wl,--no-as-needed", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([sys.executable, "-c", CHILD, shim, ROOT], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
