#!/usr/bin/env python3
"""GPU box helper: randomized differential test of the kernel paths at N = 65536 (or 32768 / 16384: fourth argument), R = 2 (or R = 4: third argument).  Every case draws a plan (on-grid,
offset, two or three classes, mixed widths, a split plan: tilings plus a remainder, or — round 5 — banks of several widths in one plan), a block count, a chunk size and a call pattern, runs it on the default
dispatch and on the spectrum-in-memory path (FDC_NO_POLY=1) and compares every output sample; every fifth case is also
compared with the oracle.  Usage: python tools/fuzz_paths.py [cases] [seed] [R] [N]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gr_fdc_amd as G  # noqa: E402
import oracle as O      # noqa: E402  (checker)

N, R = (int(sys.argv[4]) if len(sys.argv) > 4 else 65536), (int(sys.argv[3]) if len(sys.argv) > 3 else 2)
H = N - N // R
N1 = N // 256                                      # slots of the 256-bin grid
TOL = 1e-5


def rel(a, b):
    d = np.abs(a.astype(np.complex128) - b.astype(np.complex128)).max()
    return float(d / max(np.abs(b).max(), 1e-30))


def draw_plan(rng):
    kind = rng.integers(0, 11)
    if kind >= 8:                                     # round 5: banks of DIFFERENT widths in one plan (one block-kernel launch each), duplicates of a slice
        plan, names = [], []                          # (computed once, copied) and, sometimes, a remainder of channels that fit no bank
        for _ in range(int(rng.integers(2, 5))):
            L = int(rng.choice([64, 128, 256, 256, 512, 1024]))
            r = int(rng.integers(0, 256)) if L == 256 else int(rng.choice([0, L // 2] + ([L // 4, 3 * L // 4] if L <= 128 else [])))
            win = [(0.88, 1.0), (0.7, 0.9)][int(rng.integers(0, 2))]
            nslot = N // L - (1 if r else 0)
            slots = rng.permutation(nslot)[:max(1, int(rng.integers(nslot // 3, nslot + 1)))]
            plan += [(L * int(c) + r, L) + win for c in slots]
            names.append("%d@%d" % (L, r))
        for _ in range(int(rng.integers(0, 4))):      # the same slice again
            plan.append(plan[int(rng.integers(0, len(plan)))])
        if kind == 10:
            for _ in range(int(rng.integers(1, 6))):
                l = int(2 ** rng.integers(5, 12))
                plan.append((int(rng.integers(0, N - l + 1)) | 1, l, 0.7, 0.9))
        order = rng.permutation(len(plan))
        return [plan[int(i)] for i in order], "banks " + "+".join(names)
    if kind >= 6:                                     # uniform banks of another width on its grid: 1024, 512, 128, 64 (block kernels), others (spectrum path)
        L = int([512, 128, 1024, 1024, 64, 2048][int(rng.integers(0, 6))])
        half = int(rng.choice([L // 2, L // 4, 3 * L // 4])) if rng.integers(0, 3) == 0 else 0      # a bank off its grid by a multiple of a quarter channel (narrow kernel: all; 512 / 1024: half only)
        slots = rng.permutation(N // L - (1 if half else 0))[:rng.integers(1, N // L + (0 if half else 1))]
        plan = [(L * int(c) + half, L, 0.88, 1.0) for c in slots]
        two = ""
        if rng.integers(0, 3) == 0:                          # a few channels on the OTHER grid: two banks, two launches (or the spectrum path by the cost rule)
            other = int(rng.choice([r_ for r_ in (0, L // 4, L // 2, 3 * L // 4) if r_ != half]))
            oslots = rng.permutation(N // L - (1 if other else 0))[:rng.integers(1, 5)]
            plan += [(L * int(c) + other, L, 0.88, 1.0) for c in oslots]
            plan = [plan[int(i)] for i in rng.permutation(len(plan))]
            two = " x2"
        return plan, "bank l=%d%s%s" % (L, "+l/2" if half else "", two)
    if kind == 0:                                     # on-grid subset
        slots = rng.permutation(N1)[:rng.integers(1, N1 + 1)]
        return [(256 * int(c), 256, 0.88, 1.0) for c in slots], "grid"
    if kind == 1:                                     # one offset
        r = int(rng.integers(1, 256))
        slots = rng.permutation(N1 - 1)[:rng.integers(1, N1)]
        return [(256 * int(c) + r, 256, 0.88, 1.0) for c in slots], "offset %d" % r
    if kind == 2:                                     # two or three classes, enough channels for the launches to pay
        ncl = int(rng.integers(2, 4))
        plan = []
        for k in range(ncl):
            r = int(rng.integers(0, 256))
            win = [(0.88, 1.0), (0.7, 0.9), (0.8, 0.95)][int(rng.integers(0, 3))]
            slots = rng.permutation(N1 - 1)[:rng.integers(150 * N1 // 256, N1)]
            plan += [(256 * int(c) + r, 256) + win for c in slots]
        order = rng.permutation(len(plan))
        return [plan[int(i)] for i in order], "%d classes" % ncl
    if kind == 3:                                     # mixed widths: spectrum path either way, register channel kernels
        plan = []
        for _ in range(int(rng.integers(1, 12))):
            l = int(2 ** rng.integers(6, 12))
            plan.append((int(rng.integers(0, N - l + 1)), l, 0.88, 1.0))
        return plan, "mixed"
    if kind == 5:                                     # split plans: one to three tilings plus a remainder of other widths / further tilings
        plan = []
        for k in range(int(rng.integers(1, 4))):
            r = int(rng.integers(0, 256)) if k else 0
            slots = rng.permutation(N1 - 1)[:rng.integers(120 * N1 // 256, N1)]
            plan += [(256 * int(c) + r, 256, 0.88, 1.0) for c in slots]
        for _ in range(int(rng.integers(1, 9))):
            l = int(2 ** rng.integers(6, 12))
            plan.append((int(rng.integers(0, N - l + 1)), l, 0.7, 0.9))
        order = rng.permutation(len(plan))
        return [plan[int(i)] for i in order], "split"
    slots = rng.permutation(N1)[:rng.integers(1, max(2, 40 * N1 // 256))]   # few channels
    return [(256 * int(c), 256, 0.88, 1.0) for c in slots], "few"


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    worst = 0.0
    for case in range(cases):
        plan, what = draw_plan(rng)
        nb = int(rng.choice([1, 2, 3, 7, 40, 95, 96, 97, 130, 256, 300]))
        chunk = int(rng.choice([0, 0, 64, 96, 128]))
        x = (rng.standard_normal(nb * H) + 1j * rng.standard_normal(nb * H)).astype(np.complex64)
        cuts = sorted(set([0, nb] + [int(v) for v in rng.integers(0, nb + 1, size=int(rng.integers(0, 3)))]))
        if rng.integers(0, 2):
            G.defaults["FDC_BLOCK_MIN_BLOCKS"] = "1"
        else:
            G.defaults.pop("FDC_BLOCK_MIN_BLOCKS", None)
        # every other multi-width case keeps ALL its banks whatever the cost rule says (FDC_PIPE_WIDE_UNIFORM): the launches are what is tested
        flags = G.FDC_PIPE_WIDE_UNIFORM if what.startswith("banks") and rng.integers(0, 2) else None
        p = G.Pipeline(N, R, plan, windowtype=1, max_blocks=nb, chunk_blocks=chunk, flags=flags)
        path = p.path()
        parts = [p.work(x[a * H:b * H]) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
        outs = [np.concatenate([q[c] for q in parts]) for c in range(len(plan))]
        G.defaults["FDC_NO_POLY"] = "1"
        try:
            ref = G.Pipeline(N, R, plan, windowtype=1, max_blocks=nb).work(x)
        finally:
            del G.defaults["FDC_NO_POLY"]
        e = max(rel(a, b) for a, b in zip(outs, ref))
        eo = 0.0
        if case % 5 == 0 or path == 1:                # the comparison above is vacuous when both runs take the spectrum path
            oref, _ = O.channelizer(N, R, 1, plan, x, nthreads=8)
            eo = max(rel(a, b) for a, b in zip(outs, oref))
        worst = max(worst, e, eo)
        flag = "" if max(e, eo) <= TOL else "   <-- FAIL"
        print("case %3d  %-10s path %d  %3d ch  %3d blocks  chunk %3d  cuts %-14s vs spectrum path %.2e  vs oracle %.2e%s  [%s]"
              % (case, what[:40], path, len(plan), nb, chunk, cuts, e, eo, flag, p.describe().split("path ")[1][:110]), flush=True)
        if flag:
            sys.exit(1)
    print("all %d cases within %.0e (worst %.2e)" % (cases, TOL, worst))


if __name__ == "__main__":
    main()
