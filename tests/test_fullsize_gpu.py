"""GPU tests (-m gpu) at BASELINE.json's FULL sizes (cfg2 = 65536-pt FFT, 256 channels, 1024 blocks per
launch; cfg4's per-GPU shape = 262144-pt FFT, 1024 channels, 256 blocks), through the C-ABI's host entry:
  * the whole batch against the oracle (its OpenMP double-precision form finishes these sizes in seconds),
  * size-independent properties: launch grouping is invisible (bit-exact), linearity, block-shift invariance
    (bit-exact), unit gain and phase continuity of a bin-centred tone across every block boundary."""
import os

import numpy as np
import pytest

import gr_fdc_amd as G

pytestmark = pytest.mark.gpu
TOL = 1e-5


def plan_for(N, R, C):
    params = [G.get_opt_channelparams(N, R, ((c + 0.5) / C) % 1.0, 0.8 / C) for c in range(C)]
    assert all(p[:3] == (256 * c, 256, 256 - 256 // R) for c, p in enumerate(params))
    return [(f, l, p, s) for (f, l, _lo, p, s) in params]


def noise(n, seed):
    rng = np.random.default_rng(seed)
    x = np.empty(n, np.complex64)
    x.real = rng.standard_normal(n, dtype=np.float32)
    x.imag = rng.standard_normal(n, dtype=np.float32)
    return x


def bits(a):
    return a.view(np.uint32)


FORCED = any(G.defaults.get(k) for k in ("FDC_FORCE_GENERIC", "FDC_NO_POLY", "FDC_NO_BLOCK"))     # the suite itself run under a forced path


def run(N, R, plan, x, nb, sub=None, force=None):
    """force: None = the default path (uniform plan: one kernel at N = 65536, two launches otherwise), "FDC_NO_BLOCK" = the
    two-launch uniform path, "FDC_NO_POLY" = spectrum in memory, "FDC_FORCE_GENERIC" = generic kernels"""
    if sub is not None:
        G.defaults["FDC_HOST_SUB"] = str(sub)
    if force:
        G.defaults[force] = "1"
    try:
        p = G.Pipeline(N, R, plan, windowtype=1, max_blocks=nb)
        if not FORCED:
            assert p.path() == {None: 3 if N == 65536 and R in (2, 4) else 2, "FDC_NO_BLOCK": 2,
                                "FDC_NO_POLY": 1 if N == 65536 else 0, "FDC_FORCE_GENERIC": 0}[force]
        return p.work(x)
    finally:
        G.defaults.pop("FDC_HOST_SUB", None)
        if force:
            G.defaults.pop(force, None)


@pytest.mark.parametrize("N,C,nb", [(65536, 256, 1024), (262144, 1024, 256)])
def test_full_size_batch_vs_oracle_and_launch_grouping(oracle, N, C, nb):
    R = 2
    H = N - N // R
    plan = plan_for(N, R, C)
    x = noise(nb * H, N)
    outs = run(N, R, plan, x, nb)                      # sub-batches of ~8 MiB, overlapped transfers
    whole = run(N, R, plan, x, nb, sub=nb)             # one launch group for the whole batch
    for c in range(C):
        assert np.array_equal(bits(outs[c]), bits(whole[c])), "launch grouping changed channel %d" % c
    ref, _ = oracle.channelizer(N, R, 1, plan, x, nthreads=min(16, os.cpu_count() or 1))
    worst_l2 = worst_mx = 0.0
    for c in range(C):
        d = outs[c].astype(np.complex128) - ref[c].astype(np.complex128)
        worst_l2 = max(worst_l2, float(np.linalg.norm(d) / np.linalg.norm(ref[c])))
        worst_mx = max(worst_mx, float(np.abs(d).max() / np.abs(ref[c]).max()))
    assert worst_l2 <= TOL and worst_mx <= TOL, (worst_l2, worst_mx)
    # the other two kernel paths on the same full-size batch (spectrum in memory; generic kernels) agree with the oracle too
    for force in ("FDC_NO_BLOCK", "FDC_NO_POLY", "FDC_FORCE_GENERIC"):
        if force == "FDC_NO_BLOCK" and N != 65536:
            continue
        alt = run(N, R, plan, x, nb, force=force)
        for c in range(0, C, 7):
            d = alt[c].astype(np.complex128) - ref[c].astype(np.complex128)
            assert np.linalg.norm(d) <= TOL * np.linalg.norm(ref[c]) and np.abs(d).max() <= TOL * np.abs(ref[c]).max(), (force, c)


def test_bench_launch_size_one_group_of_2048_blocks(oracle):
    """bench.py's default step: 2048 blocks of configs[1] in ONE launch group of the one-kernel path (the library cuts calls into
    groups of chunk_blocks; at this size the group is the whole call).  Every output sample against the oracle, and bit for bit
    against the same stream run in sub-batches."""
    N, R, C, nb = 65536, 2, 256, 2048
    H = N - N // R
    plan = plan_for(N, R, C)
    x = noise(nb * H, 77)
    whole = run(N, R, plan, x, nb, sub=nb)
    p = G.Pipeline(N, R, plan, windowtype=1, max_blocks=nb)
    assert p.chunk_blocks() >= nb and (FORCED or p.path() == 3)
    p.close()
    pieces = run(N, R, plan, x, nb)
    for c in range(C):
        assert np.array_equal(bits(pieces[c]), bits(whole[c])), "launch grouping changed channel %d" % c
    del pieces
    ref, _ = oracle.channelizer(N, R, 1, plan, x, nthreads=min(16, os.cpu_count() or 1))
    worst_l2 = worst_mx = 0.0
    for c in range(C):
        d = whole[c].astype(np.complex128) - ref[c].astype(np.complex128)
        worst_l2 = max(worst_l2, float(np.linalg.norm(d) / np.linalg.norm(ref[c])))
        worst_mx = max(worst_mx, float(np.abs(d).max() / np.abs(ref[c]).max()))
    assert worst_l2 <= TOL and worst_mx <= TOL, (worst_l2, worst_mx)


def test_full_size_linearity_and_block_shift():
    N, R, C, nb = 65536, 2, 256, 1024
    H = N - N // R
    plan = plan_for(N, R, C)
    x1, x2 = noise(nb * H, 1), noise(nb * H, 2)
    y1, y2, y12 = run(N, R, plan, x1, nb), run(N, R, plan, x2, nb), run(N, R, plan, x1 + x2, nb)
    for c in range(0, C, 5):
        s = y1[c].astype(np.complex128) + y2[c].astype(np.complex128)
        d = y12[c].astype(np.complex128) - s
        assert np.linalg.norm(d) <= TOL * np.linalg.norm(s) and np.abs(d).max() <= TOL * np.abs(s).max()
    # drop the first block of the stream: every later block is the same block one index earlier (all f are multiples
    # of R, so the window phase does not depend on the index); only the new first block sees a different history
    ys = run(N, R, plan, x1[H:], nb - 1)
    lout = 128
    for c in range(C):
        assert np.array_equal(bits(ys[c][lout:]), bits(y1[c][2 * lout:])), "block shift changed channel %d" % c


def test_full_size_tone_gain_and_phase_continuity():
    N, R, C, nb = 65536, 2, 256, 1024
    H = N - N // R
    plan = plan_for(N, R, C)
    c0 = 77
    k_shifted = 256 * c0 + 128 + 9                      # a bin inside channel c0's pass band (centre + 9)
    k = (k_shifted - N // 2) % N                        # unshifted DFT bin
    n = np.arange(nb * H + N // R, dtype=np.int64)      # the stream including the zero-history region's successor
    ph = (k * n) % N
    tone = np.exp(2j * np.pi * ph / N).astype(np.complex64)
    y = run(N, R, plan, tone[N // R:], nb)              # history is zeros: the first block is a partial tone
    lout = 128
    z = y[c0][lout:].astype(np.complex128)              # from the second block on, every block holds the full tone
    assert np.abs(np.abs(z) - 1.0).max() < 1e-4         # net gain exactly 1 in the pass band
    step = z[1:] * np.conj(z[:-1])
    expect = np.exp(2j * np.pi * 9 / 256)               # decimated tone: 9 bins off the channel centre, l = 256
    assert np.abs(step - expect).max() < 2e-4           # same phase step inside blocks and across all 1022 boundaries
    others = max(float(np.abs(y[c][lout:]).max()) for c in range(C) if c != c0)
    assert others < 1e-4


# ---------------------------------------------------------------- cfg3 / cfg5 at full size
from test_sinks_gpu import compare  # noqa: E402  (same PDU comparison as the small cases)


def bursty_stream(N, R, nb, carriers, seed, tone_amp, seg_blocks=8, floor=0.016):
    """carriers: (first_even_bin, last_even_bin) in UNSHIFTED N-point bins.  Every carrier is a set of tones on even
    bins (periodic in H = N/2, so a segment of seg_blocks blocks is one IFFT tiled) with fixed random phases, keyed on
    and off per segment: phase-continuous.  Levels follow SURVEY.md section 8d (a carrier about 30 dB above the white floor in
    its own band): with a far larger on/off ratio the payloads that hold only floor or the splatter of a switching
    neighbour sit below what a float32 forward transform of the whole band resolves to 1e-5 of THEIR level."""
    H = N - N // R
    assert R == 2
    g = np.random.default_rng(seed)
    nseg = nb // seg_blocks
    phases = [np.exp(2j * np.pi * g.random((hi - lo) // 2 + 1)) for lo, hi in carriers]
    on = g.integers(0, 2, size=(len(carriers), nseg)).astype(bool)
    on[:, 0] = False                                      # the stream starts idle
    x = np.empty(nb * H, np.complex64)
    for s in range(nseg):
        S = np.zeros(H, np.complex128)
        for k, (lo, hi) in enumerate(carriers):
            if on[k, s]:
                b = (np.arange(lo, hi + 1, 2) // 2) % H
                S[b] = phases[k] * tone_amp
        seg = np.fft.ifft(S) * H
        x[s * seg_blocks * H:(s + 1) * seg_blocks * H] = np.tile(seg, seg_blocks).astype(np.complex64)
    x += floor * noise(nb * H, seed + 1)
    return x, on


def test_cfg3_full_size_256_power_activation_channels(oracle):
    N, R, C, nb = 65536, 2, 256, 1024
    pac = [(((c + 0.5) / C) % 1.0, 0.8 / C, c) for c in range(C)]
    carriers = []
    for c in range(C):                                    # a 40-bin multitone at the centre of every channel
        k = (256 * c + 128 - N // 2) % N
        carriers.append((k - 20, k + 20))
    x, on = bursty_stream(N, R, nb, carriers, 2026, tone_amp=8.7e-3)      # 21 tones: 30 dB over the in-band floor
    p = G.Pipeline(N, R, [], windowtype=1, max_blocks=nb, keep_spectrum=True)
    bank = G.Sinks(N, R, pac=pac, pac_thresh=6.0, pac_maxblocks=128, pac_delay=1, max_blocks=nb)
    p.work(x, sinks=bank)
    got = bank.pdus()
    _, spec = oracle.channelizer(N, R, 1, [], x, want_spectrum=True, nthreads=min(16, os.cpu_count() or 1))
    spec = spec.reshape(nb, N)
    total = 0
    for c in range(0, C):
        ref = oracle.PowerActivationChannel(N, pac[c][0], pac[c][1], R, 6.0, 128, 1, c).work(spec)
        mine = [g for g in got if g[0]["source"] == c]
        compare(mine, ref, vec=False)
        total += len(ref)
    assert total == len(got) and total > C                # every channel was active at least once on average


def test_cfg5_full_size_activity_detection(oracle):
    N, R, nb = 65536, 2, 1024
    segs = [[0.05, 0.45], [0.55, 0.95]]
    g = np.random.default_rng(2028)
    carriers = []
    for s0, s1 in segs:                                   # 12 carriers per segment, widths 0.002-0.03, non-overlapping
        pos = s0 + 0.01
        while pos < s1 - 0.04 and len(carriers) < 12 * (1 + segs.index([s0, s1])):
            w = float(g.uniform(0.002, 0.03))
            lo, hi = int(pos * N), int((pos + w) * N)
            lo_u, hi_u = (lo - N // 2) % N, (hi - N // 2) % N
            if lo_u < hi_u:
                carriers.append((lo_u + (lo_u & 1), hi_u - (hi_u & 1)))
            pos += w + float(g.uniform(0.01, 0.02))
    x, on = bursty_stream(N, R, nb, carriers, 2029, tone_amp=4e-3)        # density 30 dB over the floor
    p = G.Pipeline(N, R, [], windowtype=1, max_blocks=nb, keep_spectrum=True)
    det = G.Sinks(N, R, segments=[tuple(s) for s in segs], det_thresh=10.0, det_maxblocks=128, minchandist=0.005,
                  det_delay=1, puffer=0.2, max_blocks=nb)
    assert det.segment_params(0)["dec"] == 163
    p.work(x, sinks=det)
    got = det.pdus()
    _, spec = oracle.channelizer(N, R, 1, [], x, want_spectrum=True, nthreads=min(16, os.cpu_count() or 1))
    ref = oracle.ActivityDetectionVcm(N, segs, 10.0, R, 128, 0.005, 1, 0.2).work(spec.reshape(nb, N))
    assert len(ref) > 50
    compare(got, ref)


def test_combined_bank_both_threaded_phases_two_calls(oracle):
    """One bank with 64 PowerActivationChannels AND two detection segments, fed in two calls of 256 blocks: both decision
    phases run on the worker threads (PowerActivationChannels by channel, segments by segment), their lists are merged twice
    per call, PDUs that span the call boundary carry blocks over, and the emission order is restored from the order keys.
    Every PDU against the oracle; the order inside a call: block-major, PowerActivationChannels before detections."""
    N, R, nb = 65536, 2, 512
    C = 64
    pac = [(((4 * c + 0.5) / 256) % 1.0, 0.8 / 256, c) for c in range(C)]
    carriers = []
    for c in range(C):
        k = (256 * 4 * c + 128 - N // 2) % N
        carriers.append((k - 20, k + 20))
    x, _ = bursty_stream(N, R, nb, carriers, 77, tone_amp=8.7e-3)
    segs = [[0.05, 0.45], [0.55, 0.95]]
    p = G.Pipeline(N, R, [], windowtype=1, max_blocks=256, keep_spectrum=True)
    bank = G.Sinks(N, R, pac=pac, pac_thresh=6.0, pac_maxblocks=128, pac_delay=1, segments=[tuple(s) for s in segs],
                   det_thresh=10.0, det_maxblocks=128, minchandist=0.005, det_delay=1, puffer=0.2, max_blocks=256)
    H = N - N // R
    got = []
    for a in (0, 256):
        p.work(x[a * H:(a + 256) * H], sinks=bank)
        pdus = bank.pdus()
        keys = [(int(m["blockend"]), 0 if m["kind"] == 0 else 1) for m, _ in pdus]
        assert keys == sorted(keys), "emission order inside a call"
        got += pdus
    _, spec = oracle.channelizer(N, R, 1, [], x, want_spectrum=True, nthreads=min(16, os.cpu_count() or 1))
    spec = spec.reshape(nb, N)
    npac = 0
    for c in range(C):
        ref = oracle.PowerActivationChannel(N, pac[c][0], pac[c][1], R, 6.0, 128, 1, c).work(spec)
        compare([g for g in got if g[0]["kind"] == 0 and g[0]["source"] == c], ref, vec=False)
        npac += len(ref)
    ref = oracle.ActivityDetectionVcm(N, segs, 10.0, R, 128, 0.005, 1, 0.2).work(spec)
    compare([g for g in got if g[0]["kind"] == 1], ref)
    assert npac > C and len(ref) > 10 and npac + len(ref) == len(got)
