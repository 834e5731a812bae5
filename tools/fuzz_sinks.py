#!/usr/bin/env python3
"""GPU box helper: randomized differential test of the two engines of a sink bank (FDC_SINKS_*): every case draws a bank
(PowerActivationChannels of several widths and / or detection segments, vcm or SegmentDetection form), thresholds, maxblocks,
deactivation delay, flank puffer, a spectrum of carriers keyed on and off at random — including carriers that touch, merge and
split, zero-power stretches and carriers wider than a block after the puffer — and a pattern of calls, and runs it on the device
engine and on the host engine.  Everything must agree exactly: order, metadata, payload.
Usage: python tools/fuzz_sinks.py [cases] [seed] [max blocks per case, default 90]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gr_fdc_amd as G  # noqa: E402

KEYS = ("kind", "source", "chan_id", "finalized", "part", "has_part", "blockstart", "blockend", "vectorstart", "vectorend",
        "rel_bw", "rel_cfreq")


def spectrum(rng, N, nb, carriers, floor, zero_prob):
    s = floor * (rng.standard_normal((nb, N)) + 1j * rng.standard_normal((nb, N)))
    for lo, hi, amp in carriers:
        m, on = 0, bool(rng.integers(0, 2))
        while m < nb:
            ln = int(rng.integers(1, 12))
            if on:
                k = min(nb, m + ln) - m
                s[m:m + ln, lo:hi] += amp * (rng.standard_normal((k, hi - lo)) + 1j * rng.standard_normal((k, hi - lo)))
            m += ln
            on = not on
    if zero_prob > 0:                                  # stretches of exact zeros (division guards, FLT_MIN substitution)
        for _ in range(int(rng.integers(1, 4))):
            a = int(rng.integers(0, N - 64))
            b0 = int(rng.integers(0, nb))
            s[b0:b0 + int(rng.integers(1, 6)), a:a + int(rng.integers(8, 200))] = 0
    return s.astype(np.complex64)


def draw(rng, nbmax=90):
    N = int(2 ** rng.integers(10, 15))
    R = int(2 ** rng.integers(1, 3))
    nb = int(rng.integers(8, nbmax))
    kw = dict(max_blocks=int(rng.integers(3, 40)) if nbmax <= 90 or rng.random() < 0.3 else int(rng.integers(40, nbmax + 1)))
    carriers = []
    what = int(rng.integers(0, 3))                     # 0 PAC only, 1 detection only, 2 both
    if what in (0, 2):
        npac = int(rng.integers(1, 80))
        pac = []
        for i in range(npac):
            bw = float(rng.uniform(0.001, 0.04))
            cf = float(rng.uniform(bw / 2 + 0.001, 1 - bw / 2 - 0.001))
            pac.append((cf, bw, int(rng.integers(0, 1000))))
            lo, hi = int(round((cf - bw / 2) * N)), int(round((cf + bw / 2) * N))
            if hi > lo and rng.random() < 0.8:
                carriers.append((lo, hi, float(rng.uniform(0.05, 1.0))))
        kw.update(pac=pac, pac_thresh=float(rng.uniform(1.0, 12.0)), pac_maxblocks=int(rng.choice([-1, 0, 1, 2, 3, 5, 17])))
    if what in (1, 2):
        variant = int(rng.integers(0, 2))
        if variant == 0:
            nseg = int(rng.integers(1, 4))
            edges = np.sort(rng.uniform(0.02, 0.98, 2 * nseg))
            segs = [(float(edges[2 * i]), float(edges[2 * i + 1])) for i in range(nseg) if edges[2 * i + 1] - edges[2 * i] > 0.02]
        else:
            a, b = sorted(rng.uniform(0.02, 0.98, 2))
            segs = [(float(a), float(b))] if b - a > 0.02 else [(0.1, 0.9)]
        if not segs:
            segs = [(0.1, 0.9)]
        kw.update(segments=segs, det_thresh=float(rng.uniform(3.0, 13.0)), det_maxblocks=int(rng.choice([-1, 0, 1, 2, 3, 5, 17])),
                  minchandist=float(rng.uniform(0.004, 0.05)), det_delay=int(rng.integers(0, 4)), puffer=float(rng.choice([0.0, 0.1, 0.2, 0.6])),
                  det_variant=variant)
        for (a, b) in segs:
            pos = a + 0.005
            while pos < b - 0.01:
                w = float(rng.uniform(0.003, 0.06))
                carriers.append((int(pos * N), max(int(pos * N) + 1, int(min(b, pos + w) * N)), float(rng.uniform(0.05, 1.0))))
                pos += w + float(rng.choice([0.0, 0.0, 0.01, 0.04]))      # gap 0: carriers that touch (merge / split as they key)
    spec = spectrum(rng, N, nb, carriers, float(rng.choice([1e-3, 1e-2])), float(rng.random() < 0.3))
    cuts = sorted(set(int(v) for v in rng.integers(1, nb, int(rng.integers(0, 8)))))
    return N, R, kw, spec, cuts


def run(bank, spec, cuts, max_blocks):
    out, a = [], 0
    for b in list(cuts) + [spec.shape[0]]:
        if b > a:
            out += bank.work(spec[a:b].reshape(-1))
        a = b
    return out


def main(cases=60, seed=1, nbmax=90, lookahead=0):
    """lookahead = 1 (fourth argument): the device-engine bank is a look-ahead bank (two spectrum buffers that swap per batch, the extractions on a
    stream of their own, wide classes on side streams) — fed through work() like the other one: same PDUs."""
    rng = np.random.default_rng(seed)
    npdu = ndev = 0
    for case in range(cases):
        N, R, kw, spec, cuts = draw(rng, nbmax)
        try:
            dev, host = G.Sinks(N, R, lookahead=bool(lookahead), **kw), G.Sinks(N, R, host_decisions=True, **kw)
        except ValueError:
            continue                                    # the reference's constructors refuse this geometry too
        if dev.engine() != 1:
            continue
        ndev += 1
        a, b = run(dev, spec, cuts, kw["max_blocks"]), run(host, spec, cuts, kw["max_blocks"])
        what = "case %d (seed %d): N %d R %d %s cuts %s" % (case, seed, N, R, {k: v for k, v in kw.items() if k not in ("pac", "segments")}, cuts)
        assert len(a) == len(b), (what, len(a), len(b))
        for k, ((ma, da), (mb, db)) in enumerate(zip(a, b)):
            for key in KEYS:
                assert ma[key] == mb[key], (what, k, key, ma, mb)
            assert ma["id"][19:] == mb["id"][19:], (what, k)
            assert da.size == db.size and (da == db).all(), (what, k, da.size, db.size)
        npdu += len(a)
    print("fuzz_sinks: %d cases on the device engine, %d PDUs, all equal to the host engine" % (ndev, npdu))
    return ndev, npdu


if __name__ == "__main__":
    main(*[int(v) for v in sys.argv[1:5]])
