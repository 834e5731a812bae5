#!/usr/bin/env python3
"""Generates tests/golden/channel_params.json by RUNNING the reference's own Python caller.

Runs in the build container only (needs /root/reference); never on the GPU box.  The reference file
python/FrequencyDomainChannelizer.py imports gnuradio, FDC (its SWIG module) and pmt at module level; none of
them is used by the three things pinned here, so they are replaced in sys.modules by attribute-less namespace
modules (the class statement needs `gr.hier_block2` to be *a* base class: `object`).  What is then executed is
the reference's code, unmodified and in place:

  * nextpow2                                   python/FrequencyDomainChannelizer.py:37-40
  * the freq-mode lambdas get_freq/get_bw/set_freq/set_bw and get_channel/get_segment   :70-91, :349-357
    (they are created at the top of __init__; __init__ is entered with blocksize=0 so that it leaves through
    nextpow2's ValueError at :138, after the lambdas and the converted channel lists exist and before any
    GNU Radio object would be needed)
  * get_opt_channelparams                      :322-345 (called on an instance that carries blocksize/relinvovl)

Rounding note.  The reference is Python-2 code (python/__init__.py:29,34, print statements elsewhere): its
`round()` at :336 rounds half AWAY from zero; the interpreter here is Python 3 (half to even).  Each row is
evaluated twice: "out_py3" with the builtins of this interpreter and "out" with a Python-2 `round` placed in the
imported module's namespace (the builtin of the reference's own interpreter; the reference source is not edited).
The two differ only on exact .5 ties of freq*blocksize; rows where they differ carry "tie": true.

Usage:  python3 tests/golden/make_params_from_reference.py     (rewrites channel_params.json next to it)
"""
import importlib.util
import json
import math
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/python/FrequencyDomainChannelizer.py"


def load_reference():
    for name in ("gnuradio", "gnuradio.gr", "gnuradio.blocks", "gnuradio.fft", "FDC", "pmt"):
        sys.modules.setdefault(name, types.ModuleType(name))
    gnuradio = sys.modules["gnuradio"]
    gnuradio.gr, gnuradio.blocks, gnuradio.fft = (sys.modules["gnuradio." + n] for n in ("gr", "blocks", "fft"))
    gnuradio.gr.hier_block2 = object
    for n in ("overlap_save", "phase_shifting_windowing_vcc", "vector_cut_vxx", "PowerActivationChannel", "SegmentDetection"):
        setattr(sys.modules["FDC"], n, None)
    spec = importlib.util.spec_from_file_location("ref_fdc_py", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def py2_round(x):
    """Python 2's round(): half away from zero, returns float."""
    ax = abs(x)                                  # decided on the fraction itself: abs(x) + 0.5 rounds 0.49999999999999994 up
    r = math.floor(ax)
    if ax - r >= 0.5:
        r += 1.0
    return float(r) * (1.0 if x >= 0 else -1.0)


def instance(mod, freqmode, fs, cf, channels, segments):
    """An instance whose __init__ ran up to :138 (lambdas + converted lists set)."""
    obj = object.__new__(mod.FrequencyDomainChannelizer)
    try:
        mod.FrequencyDomainChannelizer.__init__(
            obj, 8, 1, 0, 2, channels, None, 6.0, fs, cf, freqmode, 1, False, False, "", False,
            segments, 10.0, 0.005, 1, 0.2, 0, 1, 128, 128, False)
    except ValueError as e:
        assert "next power 2" in str(e), e
    else:
        raise AssertionError("__init__ was expected to leave through nextpow2(0)")
    return obj


def main():
    mod = load_reference()
    rng = np.random.default_rng(20261004)
    rows = []

    def params(N, R, u, bw, mode=0, fs=1.0, cf=0.0, note=None):
        obj = instance(mod, mode, fs, cf, [[u, bw]], None)
        fr, b = obj.throughput_channels[0]
        obj.blocksize, obj.relinvovl = mod.nextpow2(N), mod.nextpow2(R)
        res = {}
        for key, rnd in (("out_py3", None), ("out", py2_round)):
            if rnd is None:
                mod.__dict__.pop("round", None)
            else:
                mod.round = rnd
            try:
                res[key] = list(obj.get_opt_channelparams(fr, b))
            except ValueError as e:
                res[key] = {"raises": "ValueError", "msg": str(e)}
        mod.__dict__.pop("round", None)
        row = dict(N=int(N), R=int(R), freq=float(u), bw=float(bw), freqmode=int(mode), fs=float(fs),
                   centerfrequency=float(cf), internal=[float(fr), float(b)], out=res["out"], out_py3=res["out_py3"],
                   tie=res["out"] != res["out_py3"])
        if note:
            row["note"] = note
        rows.append(row)

    # the five rows SURVEY.md section 8 a8 quotes
    for (N, R, u, bw) in [(4096, 4, 0.12, 0.05), (4096, 4, 0.22, 0.1), (4096, 4, -0.14, 0.12), (4096, 4, 0.0, 0.081),
                          (65536, 2, 0.5 / 256 - 0.5, 0.8 / 256)]:
        params(N, R, u, bw, note="SURVEY a8")
    # BASELINE plans: every channel of cfg2 (256 @ 65536) and cfg4 (1024 @ 262144), cfg1's list at R = 2 and 4, +1 bin
    for c in range(256):
        params(65536, 2, (c + 0.5) / 256 - 0.5, 0.8 / 256, note="cfg2")
    for c in range(0, 1024, 7):
        params(262144, 2, (c + 0.5) / 1024 - 0.5, 0.8 / 1024, note="cfg4")
    for R in (2, 4):
        for (u, bw) in [(0.12, 0.05), (0.22, 0.1), (-0.14, 0.12), (0.0, 0.081)]:
            params(4096, R, u, bw, note="cfg1")
            params(4096, R, u + 1.0 / 4096, bw, note="cfg1 +1 bin")
    # random sweep, normalised mode
    for _ in range(700):
        N = int(2 ** rng.integers(5, 21)); R = int(2 ** rng.integers(1, 5))
        params(N, R, float(rng.uniform(-0.5, 0.5)), float(rng.uniform(1.5 / N, 0.45)))
    # exact .5 ties of freq*blocksize (round half away vs half even), both parities of the integer part
    for N in (64, 4096, 65536):
        for k in list(range(0, 12)) + [N // 2 - 3, N // 2 - 2, N - 6, N - 5]:
            params(N, 2, (k + 0.5) / N - 0.5, 4.0 / N * 3, note="tie")
    # clamp at the upper edge, wrap below zero, passband < 0.7, passband clamp to 1, l doubling boundary
    for N in (256, 4096, 65536):
        for u in (-0.5, -0.5 + 1.0 / N, -0.499, 0.4999, 0.5 - 1.0 / N, 0.49, -0.49):
            for bw in (0.26, 0.1, 0.031, 3.0 / N):
                params(N, 4, u, bw, note="edge")
        for occ in (1.0, 1.01, 2.0, 100.0, 106.0, 106.7, 107.0, 128.0, 128.5, 213.0, 213.4):
            params(N, 2, 0.1, occ / N, note="l boundary")
    params(4096, 2, 0.1, 0.6, note="wide"); params(4096, 2, 0.1, 0.95, note="wider than the band after doubling")
    params(4096, 2, 0.1, 1.0, note="bw % 1.0 == 0 -> nextpow2(0) raises"); params(4096, 2, 0.1, 0.1 / 4096, note="below one bin")
    params(1000, 3, 0.2, 0.1, note="blocksize/relinvovl rounded up to powers of two (:138-139)")
    # the two fs-scaled modes
    for _ in range(150):
        N = int(2 ** rng.integers(8, 19)); R = int(2 ** rng.integers(1, 4))
        fs = float(rng.choice([1e6, 2.4e6, 48e3, 20e6, 3.0]))
        cf = float(rng.choice([0.0, 100e6, 433.92e6, -7.5]))
        u = float(rng.uniform(-0.5, 0.5)); bw = float(rng.uniform(1.5 / N, 0.3))
        params(N, R, u * fs, bw * fs, mode=1, fs=fs, note="basebandfs")
        params(N, R, u * fs + cf, bw * fs, mode=2, fs=fs, cf=cf, note="centerfreqfs")
    # string spellings of the modes (:75-84) and segments through get_segment
    modes = []
    for mode, fs, cf in ((0, 1.0, 0.0), ("normalized", 1.0, 0.0), (1, 2e6, 0.0), ("basebandfs", 2e6, 0.0),
                         (2, 2e6, 100e6), ("centerfreqfs", 2e6, 100e6)):
        scale = 1.0 if mode in (0, "normalized") else fs
        off = cf if mode in (2, "centerfreqfs") else 0.0
        ch = [[-0.3 * scale + off, 0.05 * scale], [0.25 * scale + off, 0.01 * scale], [0.5 * scale + off, 0.2 * scale]]
        sg = [[-0.45 * scale + off, -0.05 * scale + off], [0.05 * scale + off, 0.45 * scale + off]]
        obj = instance(mod, mode, fs, cf, ch, sg)
        modes.append(dict(freqmode=mode, fs=fs, centerfrequency=cf, channels=ch, segments=sg,
                          freqmode_int=int(obj.freqmode),
                          throughput_channels=[[float(a), float(b)] for a, b in obj.throughput_channels],
                          activity_detection_segments=[[float(a), float(b)] for a, b in obj.activity_detection_segments],
                          set_freq=[float(obj.set_freq(v)) for v in (0.0, 0.25, 0.5, 0.75)],
                          set_bw=[float(obj.set_bw(v)) for v in (0.01, 0.5)],
                          get_bw_minchandist=float(obj.get_bw(0.005 * scale))))
    nextpow2 = [[float(k), int(mod.nextpow2(k))] for k in
                [1, 1.0, 1.5, 2, 3, 4, 5, 100, 127.99, 128, 128.01, 4095, 4096, 4097, 65536, 1e6, 2 ** 20 + 1]]
    try:
        mod.nextpow2(0.5)
        np2_raises = False
    except ValueError:
        np2_raises = True
    out = dict(generated_by="tests/golden/make_params_from_reference.py (imports %s)" % REF,
               python=sys.version.split()[0], rows=rows, modes=modes, nextpow2=nextpow2, nextpow2_below_one_raises=np2_raises)
    with open(os.path.join(HERE, "channel_params.json"), "w") as fh:       # one row per line, defaults left out
        fh.write('{"generated_by": %s, "python": %s,\n"rows": [\n' % (json.dumps(out["generated_by"]), json.dumps(out["python"])))
        for i, r in enumerate(rows):
            r = {k: v for k, v in r.items() if not ((k == "fs" and v == 1.0) or (k in ("centerfrequency", "freqmode") and v == 0)
                                                    or (k == "tie" and not v) or (k == "out_py3" and not r["tie"]))}
            fh.write(json.dumps(r, separators=(",", ":")) + (",\n" if i + 1 < len(rows) else "\n"))
        fh.write('],\n"modes": %s,\n"nextpow2": %s,\n"nextpow2_below_one_raises": %s}\n'
                 % (json.dumps(modes), json.dumps(nextpow2), json.dumps(np2_raises)))
    print("rows: %d (ties: %d), modes: %d" % (len(rows), sum(r["tie"] for r in rows), len(modes)))


if __name__ == "__main__":
    main()
