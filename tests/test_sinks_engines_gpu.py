"""GPU tests (-m gpu): the two engines of a sink bank (include/fdc_amd.h, FDC_SINKS_*) against each other.

The device engine runs the work() loops of PowerActivationChannel / activity_detection_channelizer_vcm / SegmentDetection as
kernels; the host engine runs the same loops on host threads.  Both use the same power-cell and extraction kernels, so
everything must agree EXACTLY: number and order of the PDUs, every metadata field, every payload sample — over randomised
on/off patterns, every maxblocks / delay regime, and call patterns that cut the streams at arbitrary places (buffered
blocks cross the calls on the device).  The oracle is not involved here; tests/test_sinks_gpu.py and the hand-derived
scenarios hold both engines against it."""
import numpy as np
import pytest

import gr_fdc_amd as G

pytestmark = pytest.mark.gpu
KEYS = ("kind", "source", "chan_id", "finalized", "part", "has_part", "blockstart", "blockend", "vectorstart", "vectorend",
        "rel_bw", "rel_cfreq")


def same(a, b, what):
    assert len(a) == len(b), (what, len(a), len(b))
    for k, ((ma, da), (mb, db)) in enumerate(zip(a, b)):
        for key in KEYS:
            assert ma[key] == mb[key], (what, k, key, ma, mb)
        assert ma["id"][19:] == mb["id"][19:], (what, k, ma["id"], mb["id"])
        assert da.size == db.size and (da == db).all(), (what, k, da.size, db.size)


def onoff_spectrum(N, nb, carriers, seed, floor=1e-3):
    """carriers: (lo_bin, hi_bin); each keyed on/off in runs of 1-9 blocks, independently"""
    rng = np.random.default_rng(seed)
    s = floor * (rng.standard_normal((nb, N)) + 1j * rng.standard_normal((nb, N)))
    for lo, hi in carriers:
        m, on = 0, bool(rng.integers(0, 2))
        while m < nb:
            ln = int(rng.integers(1, 10))
            if on:
                s[m:m + ln, lo:hi] += rng.standard_normal((min(nb, m + ln) - m, hi - lo)) + 1j * rng.standard_normal((min(nb, m + ln) - m, hi - lo))
            m += ln
            on = not on
    return s.astype(np.complex64)


def run_calls(bank, spec, cuts):
    out, a = [], 0
    for b in list(cuts) + [spec.shape[0]]:
        if b > a:
            out += bank.work(spec[a:b].reshape(-1))
        a = b
    return out


@pytest.mark.parametrize("maxblocks", [-1, 0, 1, 2, 3, 7])
def test_pac_device_engine_equals_host_engine(maxblocks):
    N, R, nb = 2048, 2, 150
    rng = np.random.default_rng(100 + maxblocks)
    plan, carriers = [], []
    for c in range(70):                                   # two waves of channels, three widths
        cf = (c + 0.5) / 72 + 0.005
        bw = (0.004, 0.008, 0.002)[c % 3]
        plan.append((cf, bw, 100 + c))
        carriers.append((int(round((cf - bw / 2) * N)), int(round((cf + bw / 2) * N))))
    spec = onoff_spectrum(N, nb, carriers, 7 + maxblocks)
    cuts = sorted(set(int(v) for v in rng.integers(1, nb, 9)))
    dev = G.Sinks(N, R, pac=plan, pac_thresh=6.0, pac_maxblocks=maxblocks, max_blocks=40)
    host = G.Sinks(N, R, pac=plan, pac_thresh=6.0, pac_maxblocks=maxblocks, max_blocks=40, host_decisions=True)
    assert dev.engine() == 1 and host.engine() == 0
    a, b = run_calls(dev, spec, cuts), run_calls(host, spec, cuts)
    assert len(b) > 200
    same(a, b, "pac maxblocks %d" % maxblocks)
    one = G.Sinks(N, R, pac=plan, pac_thresh=6.0, pac_maxblocks=maxblocks, max_blocks=nb)
    same(run_calls(one, spec, []), b, "pac maxblocks %d, one call" % maxblocks)


@pytest.mark.parametrize("variant,maxblocks,delay", [(0, -1, 0), (0, 0, 1), (0, 1, 2), (0, 3, 1), (1, -1, 1), (1, 0, 0), (1, 1, 1), (1, 3, 2)])
def test_detection_device_engine_equals_host_engine(variant, maxblocks, delay):
    N, R, nb = 4096, 2, 120
    segs = [(0.05, 0.45), (0.55, 0.95)] if variant == 0 else [(0.1, 0.9)]
    rng = np.random.default_rng(50 + 10 * variant + maxblocks + delay)
    carriers = []
    pos = 0.07
    while pos < 0.9:
        w = float(rng.uniform(0.004, 0.04))
        carriers.append((int(pos * N), int((pos + w) * N)))
        pos += w + float(rng.uniform(0.012, 0.05))
    spec = onoff_spectrum(N, nb, carriers, 21 + maxblocks)
    cuts = sorted(set(int(v) for v in rng.integers(1, nb, 7)))
    kw = dict(segments=segs, det_thresh=10.0, det_maxblocks=maxblocks, minchandist=0.005, det_delay=delay, puffer=0.2,
              max_blocks=48, det_variant=variant)
    dev, host = G.Sinks(N, R, **kw), G.Sinks(N, R, host_decisions=True, **kw)
    assert dev.engine() == 1 and host.engine() == 0
    a, b = run_calls(dev, spec, cuts), run_calls(host, spec, cuts)
    assert len(b) > 20
    same(a, b, "detection variant %d maxblocks %d delay %d" % (variant, maxblocks, delay))


def test_combined_bank_two_deep_submission_and_device_payload():
    """PowerActivationChannels and detection segments in one bank; batches submitted two deep (PDUs of batch k are read while
    batch k+1 is in flight); payloads left on the device and fetched by address."""
    import ctypes as C
    from gr_fdc_amd import _lib
    N, R, nb, per = 4096, 2, 96, 16
    plan = [((c + 0.5) / 40, 0.01, c) for c in range(40)]
    carriers = [(int(round((cf - bw / 2) * N)), int(round((cf + bw / 2) * N))) for cf, bw, _ in plan]
    spec = onoff_spectrum(N, nb, carriers, 5)
    kw = dict(pac=plan, pac_thresh=6.0, pac_maxblocks=5, segments=[(0.05, 0.45), (0.55, 0.95)], det_thresh=10.0, det_maxblocks=4,
              minchandist=0.005, det_delay=1, puffer=0.2, max_blocks=per)
    ref = run_calls(G.Sinks(N, R, host_decisions=True, **kw), spec, range(per, nb, per))
    assert len(ref) > 50

    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    for devpay in (False, True):
        bank = G.Sinks(N, R, device_payload=devpay, **kw)
        got = []

        def take(pdus):
            for m, d in pdus:
                if devpay:
                    ptr, n = d
                    arr = np.zeros(n, np.complex64)
                    if n:
                        assert hip.hipMemcpy(arr.ctypes.data, ptr, 8 * n, 2) == 0
                    d = arr
                got.append((m, d))
        for a in range(0, nb, per):
            blk = np.ascontiguousarray(spec[a:a + per].reshape(-1))
            assert hip.hipMemcpy(bank.spectrum_ptr(), blk.ctypes.data, blk.nbytes, 1) == 0
            take(bank.submit_device(per))
        take(bank.flush())
        assert bank.flush() == []
        same(got, ref, "two-deep, device payload %s" % devpay)
        # a synchronous call in between is refused while a batch is in flight
        assert hip.hipMemcpy(bank.spectrum_ptr(), blk.ctypes.data, blk.nbytes, 1) == 0
        bank.submit_device(per)
        with pytest.raises(_lib.FdcError):
            bank.work_device(per)
        bank.flush()


def test_randomised_banks_device_engine_equals_host_engine():
    """tools/fuzz_sinks.py (random banks, thresholds, maxblocks, delays, puffers, touching / merging carriers, zero-power stretches,
    call patterns) for 30 cases, in this process: both engines, everything equal."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fuzz_sinks", os.path.join(root, "tools", "fuzz_sinks.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ndev, npdu = mod.main(30, 7)
    assert ndev >= 20 and npdu > 1000
