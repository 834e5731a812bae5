#!/usr/bin/env python3
"""GPU box helper: randomised differential test of the one-launch form of N = 4096 (csrc/fdc_fused4096.hip, fdc_pipeline_path() = 5).  Every case draws a plan
of 16- ... 1024-bin channels (any offsets, overlapping and repeated slices, two windows; sometimes a width without a row form, which must send the plan
to the spectrum path), an overlap R, a window type, a call pattern and a launch-group size, runs it on dispatch and under FDC_PIPE_NO_FUSED and compares every
output sample; every fourth case also against the oracle.  Usage: python tools/fuzz_fused4096.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gr_fdc_amd as G  # noqa: E402
import oracle as O      # noqa: E402  (checker)

N, TOL = 4096, 1e-5


def rel(a, b):
    d = np.abs(a.astype(np.complex128) - b.astype(np.complex128)).max()
    return float(d / max(np.abs(b).max(), 1e-30))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    worst, fused, samples = 0.0, 0, 0
    for case in range(cases):
        R = int(rng.choice([2, 2, 4, 4, 8, 16]))
        H = N - N // R
        plan, left = [], int(rng.choice([1024, 2048, 4096, 4096, 5120]))
        while left >= 16 and len(plan) < 28:
            l = int(rng.choice([w for w in (16, 32, 64, 128, 256, 256, 512, 1024) if w <= left]))
            win = [(0.88, 1.0), (0.6, 0.85)][int(rng.integers(0, 2))]
            plan.append((int(rng.integers(0, N - l + 1)), l) + win)
            left -= l
            if rng.random() < 0.1:
                break
        if rng.random() < 0.2:
            plan.append(plan[int(rng.integers(0, len(plan)))])                  # the same slice again
        odd = rng.random() < 0.12
        if odd:
            lo = int(rng.choice([4, 8, 2048]))
            plan.append((int(rng.integers(0, N - lo + 1)), lo, 0.8, 1.0))     # no row form: the whole plan on the spectrum path
        wt = int(rng.integers(0, 3))
        sizes = [int(v) for v in rng.integers(1, 70, size=int(rng.integers(1, 5)))]
        chunk = int(rng.choice([0, 0, 7, 32]))
        p = G.Pipeline(N, R, plan, windowtype=wt, max_blocks=max(sizes), chunk_blocks=chunk)
        q = G.Pipeline(N, R, plan, windowtype=wt, max_blocks=max(sizes), chunk_blocks=chunk, flags=G.FDC_PIPE_NO_FUSED)
        bins = sum(c[1] for c in plan)
        n = {w: sum(1 for c in plan if c[1] == w) for w in (1024, 512, 256, 128, 64, 32, 16)}
        fits = n[1024] + (n[512] + 1) // 2 + sum((n[w] + 3) // 4 for w in (256, 128, 64, 32, 16)) <= 8 and \
            1056 * n[1024] + 513 * n[512] + 272 * n[256] + 136 * n[128] + 68 * n[64] + 34 * n[32] + 17 * n[16] <= 4352
        if p.path() == 5:
            fused += 1
            assert not odd and bins >= 512 and fits, (plan, p.describe())
        else:
            assert odd or bins < 512 or not fits, (plan, p.describe())
        assert q.path() != 5
        x = (rng.standard_normal(sum(sizes) * H) + 1j * rng.standard_normal(sum(sizes) * H)).astype(np.complex64)
        got, other = [[] for _ in plan], [[] for _ in plan]
        at = 0
        for n in sizes:
            for c, (a, b) in enumerate(zip(p.work(x[at * H:(at + n) * H]), q.work(x[at * H:(at + n) * H]))):
                got[c].append(a); other[c].append(b)
            at += n
        ref = O.channelizer(N, R, wt, plan, x)[0] if case % 4 == 0 else None
        for c in range(len(plan)):
            a, b = np.concatenate(got[c]), np.concatenate(other[c])
            e = rel(a, b)
            if ref is not None:
                e = max(e, rel(a, ref[c]))
            worst = max(worst, e)
            samples += a.size
            assert e <= TOL, "case %d R=%d wt=%d sizes=%s chunk=%d ch%d %s: %.3g (%s)" % (case, R, wt, sizes, chunk, c, plan[c], e, p.describe())
        p.close(); q.close()
    print("fuzz_fused4096: %d cases (%d on the one-launch form), %d output samples compared, worst relative error %.3g" % (cases, fused, samples, worst))


if __name__ == "__main__":
    main()
