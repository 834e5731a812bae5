"""Generates the committed fixtures in tests/golden/.  Run in the BUILD container only
(`python tests/golden/make_golden.py`); needs oracle/_ref (built from /root/reference/lib/windows.h
by oracle/Makefile) for the window tables.

Fixtures are DATA (inputs + expected outputs):
  windows_ref.npz      tables produced by the reference's own cr_win (lib/windows.h:41-78), unmodified,
                       through oracle/ref_windows_driver.cpp.
  channel_params.json  NOT made here: tests/golden/make_params_from_reference.py produces it by running the
                       reference's own get_opt_channelparams / frequency-mode lambdas.
  chain_numpy.npz      the throughput chain evaluated with numpy.fft in
                       complex128 — an implementation independent of oracle/fdc_oracle.c — on seeded
                       multicarrier input, float32-rounded at the reference's stage boundaries.
  sink_known_answers.json  PDU metadata the reference's sink blocks produced in the survey session's
                       validation run, kept as known answers for the detector restatement.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import oracle as O  # noqa: E402

WINDOW_CASES = [  # (wintype, l, passbw, stopbw, R)
    (t, l, p, s, R)
    for t in (0, 1, 2)
    for (l, p, s, R) in [(256, 0.88, 1.0, 2), (256, 0.88, 1.0, 4), (1024, 0.528, 0.778, 4),
                         (512, 0.7128, 1.0, 8), (64, 1.0, 1.0, 2), (16, 0.3, 0.55, 4), (2, 0.5, 0.75, 2),
                         (128, 0.4, 0.65, 2), (4096, 0.6, 0.85, 2)]
]


def make_windows():
    assert O.have_ref(), "oracle/_ref missing: run `make -C oracle` with /root/reference present"
    d = {}
    for i, (t, l, p, s, R) in enumerate(WINDOW_CASES):
        d["case%02d" % i] = O.ref_window(t, l, np.float32(p), np.float32(s), R)
    d["params"] = np.array(WINDOW_CASES, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "windows_ref.npz"), **d)


def multicarrier(N, chans, nsamp, seed, noise=0.1):
    """One QPSK-like carrier per channel (rect-shaped random symbols at 0.6x the channel bandwidth)
    plus complex white noise."""
    rng = np.random.default_rng(seed)
    n = np.arange(nsamp)
    x = noise * (rng.standard_normal(nsamp) + 1j * rng.standard_normal(nsamp)) / np.sqrt(2)
    for (f, l, _p, _s) in chans:
        sps = max(2, int(round(N / (0.6 * l))))
        nsym = nsamp // sps + 2
        sym = (rng.integers(0, 2, nsym) * 2 - 1) + 1j * (rng.integers(0, 2, nsym) * 2 - 1)
        base = np.repeat(sym, sps)[:nsamp] / np.sqrt(2)
        fc = (f + l / 2 - N / 2) / N
        x = x + base * np.exp(2j * np.pi * fc * n)
    return x.astype(np.complex64)


def numpy_chain(N, R, wintype, chans, x):
    """App. A.1-A.4 with numpy.fft (complex128), float32 at the reference's stage boundaries."""
    ovl = N // R
    H = N - ovl
    nb = x.size // H
    xp = np.concatenate([np.zeros(ovl, np.complex64), x])
    outs = [[] for _ in chans]
    for m in range(nb):
        blk = xp[m * H: m * H + N].astype(np.complex128)
        S = np.fft.fftshift(np.fft.fft(blk)).astype(np.complex64)
        X = (S * np.float32(1.0 / N)).astype(np.complex64)
        for ci, (f, l, p, s) in enumerate(chans):
            W = O.ref_window(wintype, l, np.float32(p), np.float32(s), R)
            cnt = (m * (f % R)) % R
            a = X[f:f + l]
            w = W[cnt]
            yr = (a.real * w.real).astype(np.float32) - (a.imag * w.imag).astype(np.float32)
            yi = (a.real * w.imag).astype(np.float32) + (a.imag * w.real).astype(np.float32)
            Y = (yr.astype(np.float32) + 1j * yi.astype(np.float32)).astype(np.complex64)
            y = (np.fft.ifft(np.fft.ifftshift(Y.astype(np.complex128))) * l).astype(np.complex64)
            lout = l - l // R
            outs[ci].append((y[l - lout:] * np.float32(l)).astype(np.complex64))
    return [np.concatenate(o) for o in outs]


def make_chain():
    d = {}
    cases = []
    # cfg1: the example flowgraph's four channels (examples/FDC_example.grc), R = 2 and R = 4,
    # plus the same plan shifted by +1 bin so f is odd (exercises the phase rotation, App. B.8)
    user = [(0.12, 0.05), (0.22, 0.1), (-0.14, 0.12), (0.0, 0.081)]
    for name, N, R, shiftbin, wt, nb in [("cfg1_R2", 4096, 2, 0, 1, 8), ("cfg1_R4_odd", 4096, 4, 1, 1, 8),
                                         ("cfg1_R2_odd_rect", 4096, 2, 1, 0, 6), ("cfg1_R8_ramp", 4096, 8, 3, 2, 6)]:
        chans = []
        for (u, bw) in user:
            f, l, lout, p, s = O.channel_params(N, R, (u + 0.5) % 1.0, bw % 1.0)
            chans.append((min(f + shiftbin, N - l), l, p, s))
        H = N - N // R
        x = multicarrier(N, chans, nb * H, 1234)
        outs = numpy_chain(N, R, wt, chans, x)
        d[name + "_x"] = x
        for i, o in enumerate(outs):
            d[name + "_out%d" % i] = o
        cases.append(dict(name=name, N=N, R=R, wintype=wt, nblocks=nb,
                          chans=[[int(c[0]), int(c[1]), float(c[2]), float(c[3])] for c in chans]))
    d["cases"] = np.array(json.dumps(cases))
    np.savez_compressed(os.path.join(HERE, "chain_numpy.npz"), **d)


def make_sink_known_answers():
    ka = {
        "source": "SURVEY.md section 8c validation run: reference sink blocks fed a synthetic burst "
                  "(bins 1600-1799 active in blocks 3-7, N=4096, R=4)",
        "PowerActivationChannel": dict(rel_bw=0.0625, blockstart=2, blockend=9, nsamples=1344),
        "SegmentDetection": dict(vectorstart=1442, vectorend=1954, rel_bw=0.125, blockstart=2, blockend=9,
                                 nsamples=2688),
        "activity_detection_channelizer_vcm": dict(vectorstart=1442, vectorend=1954, rel_bw=0.125,
                                                   blockstart=3, blockend=10, nsamples=2688),
    }
    with open(os.path.join(HERE, "sink_known_answers.json"), "w") as fh:
        json.dump(ka, fh, indent=1)


if __name__ == "__main__":
    make_windows()
        make_chain()
    make_sink_known_answers()
    print("fixtures written to", HERE)
