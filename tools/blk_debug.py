import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gr_fdc_amd as G
N, R, nb = 65536, 2, 3
H = N // 2
chans = [(256 * c, 256, 0.88, 1.0) for c in range(256)]
rng = np.random.default_rng(1)
x = (rng.standard_normal(nb * H) + 1j * rng.standard_normal(nb * H)).astype(np.complex64)
a = G.Pipeline(N, R, chans, windowtype=1, max_blocks=nb).work(x)
os.environ["FDC_NO_BLOCK"] = "1"
b = G.Pipeline(N, R, chans, windowtype=1, max_blocks=nb).work(x)
bad = []
for c in range(256):
    d = np.abs(a[c] - b[c])
    if d.max() > 1e-4 * np.abs(b[c]).max():
        rows = np.nonzero(d > 1e-4 * np.abs(b[c]).max())[0]
        bad.append((c, len(rows), rows[:6].tolist(), bool(np.all(a[c][rows] == 0))))
print(len(bad), bad[:12])
B = np.stack(b)
for c in (0, 8, 16, 1):
    if np.abs(a[c]).max() == 0:
        print("slot", c, "all zero"); continue
    for blk in range(nb):
        seg = a[c][blk * 128:(blk + 1) * 128]
        # which (slot, block, row offset) does this look like?
        best = None
        for c2 in range(256):
            for b2 in range(nb):
                ref = B[c2][b2 * 128:(b2 + 1) * 128]
                e = np.abs(seg - ref).max()
                if best is None or e < best[0]:
                    best = (e, c2, b2)
        print("slot", c, "block", blk, "nonzero rows", int((np.abs(seg) > 0).sum()), "closest ref (err, slot, block)", best)
nz = [c for c in range(256) if np.abs(a[c]).max() > 0]
print("slots with any data:", nz[:40])
