"""GPU parity tests (-m gpu) of the one-launch form of N = 4096 (csrc/fdc_fused4096.hip, fdc_pipeline_path() = 5): the block length of the
reference's example flowgraph (examples/FDC_example.grc) and of BASELINE configs[0].  Every case is compared with the oracle AND with the
two-launch spectrum path of the same plan (FDC_PIPE_NO_FUSED: k_fft4096 + k_c256 / k_c512 / k_c1024), whose arithmetic it repeats.

Tolerance as in test_parity_gpu.py (north_star): relative L2 error <= 1e-5 and max|y - ref| / max|ref| <= 1e-5."""
import numpy as np
import pytest

import gr_fdc_amd as G
from test_parity_gpu import assert_close, noise

pytestmark = pytest.mark.gpu
N = 4096
FORCED = any(G.defaults.get(k) for k in ("FDC_FORCE_GENERIC", "FDC_NO_POLY", "FDC_NO_FUSED"))     # the suite itself run under a forced path

EXAMPLE = [(100, 256, 0.8, 1.0), (700, 512, 0.75, 0.95), (1500, 1024, 0.8, 1.0), (3001, 512, 0.6, 0.9)]      # l = 256 / 512 / 1024 / 512, one odd offset


def plans():
    rng = np.random.default_rng(11)
    full256 = [(256 * c, 256, 0.88, 1.0) for c in range(16)]
    return {
        "example flowgraph": EXAMPLE,
        "configs[0]: four 256-bin channels": [(300 + 900 * c + c, 256, 0.8, 1.0) for c in range(4)],
        "one 512-bin channel": [(1234, 512, 0.5, 0.7)],
        "one 1024-bin channel at the band edge": [(3072, 1024, 0.9, 1.0)],
        "full band of 256-bin channels (four waves of four rows)": full256,
        "1024 + 512 + ten 256 (two waves of two sets)": [(5, 1024, 0.8, 1.0), (1111, 512, 0.8, 1.0)] + [(int(rng.integers(0, N - 255)), 256, 0.8, 1.0) for _ in range(10)],
        "three 1024 + 512 + two 256": [(0, 1024, 0.8, 1.0), (1000, 1024, 0.7, 0.9), (3072, 1024, 0.8, 1.0), (2100, 512, 0.8, 1.0), (7, 256, 0.8, 1.0), (3333, 256, 1.0, 1.0)],
        "eight 512": [(512 * c, 512, 0.8, 1.0) for c in range(8)],
        "four 1024": [(1024 * c, 1024, 0.8, 1.0) for c in range(4)],
        "five 256 (a second wave for one row)": [(700 * c + 3, 256, 0.8, 1.0) for c in range(5)],
        "narrow channels: 128 and 64 bins beside the example": EXAMPLE + [(2000, 128, 0.8, 1.0), (2101, 64, 0.7, 0.9), (37, 128, 0.5, 0.8), (4032, 64, 0.8, 1.0)],
        "down to 16 bins": EXAMPLE[:2] + [(2000, 32, 0.8, 1.0), (2101, 16, 0.7, 0.9), (3, 32, 0.5, 0.8), (4080, 16, 0.8, 1.0), (777, 16, 1.0, 1.0), (1900, 64, 0.8, 1.0)],
        "nine 32-bin and nine 16-bin channels beside a 1024-bin one": [(1000, 1024, 0.8, 1.0)] + [(int(f), 32, 0.8, 1.0) for f in rng.integers(0, N - 31, 9)] +
                                                                    [(int(f), 16, 0.6, 0.9) for f in rng.integers(0, N - 15, 9)],
        "full band of 128-bin channels": [(128 * c, 128, 0.88, 1.0) for c in range(32)],
        "thirty-two 64-bin channels": [(int(f), 64, 0.8, 1.0) for f in rng.integers(0, N - 63, 32)],
        "nine 128 + nine 64 + three 256": [(int(f), 128, 0.8, 1.0) for f in rng.integers(0, N - 127, 9)] + [(int(f), 64, 0.6, 0.9) for f in rng.integers(0, N - 63, 9)] +
                                          [(int(f), 256, 0.8, 1.0) for f in rng.integers(0, N - 255, 3)],
        "the same slice twice and overlapping slices": [(500, 256, 0.8, 1.0), (500, 256, 0.8, 1.0), (600, 256, 0.5, 0.8), (400, 512, 0.8, 1.0), (650, 512, 0.8, 1.0)],
    }


@pytest.mark.parametrize("R", [2, 4, 8, 16])
def test_one_launch_form_vs_oracle_and_two_launch_form(oracle, R):
    H = N - N // R
    for name, chans in plans().items():
        for wt, nb in ((1, 9), (2, 1), (0, 37)):
            x = noise(nb * H, nb + R)
            p = G.Pipeline(N, R, chans, windowtype=wt, max_blocks=nb)
            q = G.Pipeline(N, R, chans, windowtype=wt, max_blocks=nb, flags=G.FDC_PIPE_NO_FUSED)
            assert FORCED or (p.path() == 5 and "k_f4096" in p.describe()), (name, p.describe())
            assert q.path() in (0, 2)
            outs, other = p.work(x), q.work(x)
            ref, _ = oracle.channelizer(N, R, wt, chans, x)
            for c, (o, t, r) in enumerate(zip(outs, other, ref)):
                assert_close(o, r, "%s R=%d wt=%d nb=%d ch%d vs oracle" % (name, R, wt, nb, c))
                assert_close(o, t, "%s R=%d wt=%d nb=%d ch%d vs the two-launch form" % (name, R, wt, nb, c))


def test_state_across_ragged_calls_and_launch_groups(oracle):
    """History and the phase counter carry over calls of any length (overlap_save_impl.h:33, phase_shifting_windowing_vcc_impl.h:47); a call longer than
    the launch group (chunk_blocks) is several launches of the kernel."""
    R, wt = 4, 1
    H = N - N // R
    sizes = [1, 7, 8, 9, 3, 64, 129, 2]
    x = noise(sum(sizes) * H, 77)
    ref, _ = oracle.channelizer(N, R, wt, EXAMPLE, x)
    for chunk in (0, 16):
        p = G.Pipeline(N, R, EXAMPLE, windowtype=wt, max_blocks=max(sizes), chunk_blocks=chunk)
        assert FORCED or p.path() == 5
        got = [[] for _ in EXAMPLE]
        at = 0
        for n in sizes:
            for c, o in enumerate(p.work(x[at * H:(at + n) * H])):
                got[c].append(o)
            at += n
        for c in range(len(EXAMPLE)):
            assert_close(np.concatenate(got[c]), ref[c], "ragged calls, chunk %d, ch%d" % (chunk, c))
        p.reset()
        again = p.work(x[:sizes[0] * H])
        for c in range(len(EXAMPLE)):
            assert_close(again[c], got[c][0], "after reset ch%d" % c)


def test_a_spectrum_call_on_a_one_launch_plan(oracle):
    """A call that asks for the spectrum (debug port, python/FrequencyDomainChannelizer.py:152-158) runs the spectrum path of the same handle; the
    calls around it stay on the one-launch form and the stream state is one."""
    R, wt, nb = 2, 1, 6
    H = N - N // R
    x = noise(3 * nb * H, 3)
    p = G.Pipeline(N, R, EXAMPLE, windowtype=wt, max_blocks=nb, keep_spectrum=True)
    ref, rspec = oracle.channelizer(N, R, wt, EXAMPLE, x, want_spectrum=True)
    a = p.work(x[:nb * H])
    b, spec = p.work(x[nb * H:2 * nb * H], want_spectrum=True)
    c = p.work(x[2 * nb * H:])
    assert_close(np.asarray(spec).reshape(-1), rspec[nb * N:2 * nb * N], "spectrum of the middle call")
    for ch in range(len(EXAMPLE)):
        assert_close(np.concatenate([a[ch], b[ch], c[ch]]), ref[ch], "ch%d" % ch)


def test_full_size_launch_round_trip_properties():
    """2048 blocks in one launch: linearity and block independence — the outputs of a long call equal those of the same blocks in short calls,
    on every block (the XCD-wise block mapping of the kernel puts neighbours of a launch on different workgroup ids)."""
    R, wt, nb = 2, 1, 2048
    H = N - N // R
    chans = plans()["configs[0]: four 256-bin channels"]
    x = noise(nb * H, 5)
    p = G.Pipeline(N, R, chans, windowtype=wt, max_blocks=nb)
    whole = p.work(x)
    p.reset()
    parts = [p.work(x[i * H:(i + 250) * H]) for i in range(0, nb, 250)]
    for c in range(len(chans)):
        assert np.array_equal(np.concatenate([q[c] for q in parts]), whole[c]), "ch%d: long call != short calls" % c
    p.reset()
    twice = p.work((2 * x).astype(np.complex64))
    for c in range(len(chans)):
        assert np.array_equal(twice[c], 2 * whole[c])
