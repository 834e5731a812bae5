"""CPU tests (-m "not gpu"): the oracle (oracle/fdc_oracle.c) against the committed fixtures."""
import json
import os

import numpy as np
import pytest


def rel_err(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b), np.abs(a - b).max() / np.abs(b).max()


def test_window_tables_bit_exact_vs_reference_fixture(oracle, golden_dir):
    """fdco_window == the reference's own cr_win output (lib/windows.h), bit for bit."""
    z = np.load(os.path.join(golden_dir, "windows_ref.npz"))
    params = z["params"]
    for i, (t, l, p, s, R) in enumerate(params):
        w = oracle.window(int(t), int(l), np.float32(p), np.float32(s), int(R))
        ref = z["case%02d" % i]
        assert w.shape == ref.shape
        assert (w.view(np.uint32) == ref.view(np.uint32)).all(), "case %d" % i


@pytest.mark.skipif(not os.path.exists("/root/reference/lib/windows.h"), reason="reference not mounted")
def test_window_tables_vs_live_reference_build(oracle):
    """When /root/reference is present: compare with oracle/_ref built from it, more parameter sets."""
    rng = np.random.default_rng(7)
    for _ in range(60):
        t = int(rng.integers(0, 3)); l = int(2 ** rng.integers(1, 12)); R = int(2 ** rng.integers(1, 4))
        p = np.float32(rng.uniform(0.05, 1.1)); s = np.float32(p + rng.uniform(0, 0.5))
        a = oracle.window(t, l, p, s, R); b = oracle.ref_window(t, l, p, s, R)
        assert (a.view(np.uint32) == b.view(np.uint32)).all(), (t, l, p, s, R)


def test_channel_params_vs_reference_fixture(oracle, golden_dir):
    """get_opt_channelparams restatement == the reference's own function (python/FrequencyDomainChannelizer.py:322-345),
    on the rows tests/golden/make_params_from_reference.py recorded by running it (Python-2 rounding, ties marked)."""
    fx = json.load(open(os.path.join(golden_dir, "channel_params.json")))
    n = 0
    for r in fx["rows"]:
        fr, bw = r["internal"]
        N, R = 1 << (r["N"] - 1).bit_length(), 1 << (r["R"] - 1).bit_length()
        if isinstance(r["out"], dict):
            with pytest.raises(ValueError):
                oracle.channel_params(N, R, fr, bw)
            continue
        assert list(oracle.channel_params(N, R, fr, bw)) == r["out"], r
        n += 1
    assert n >= 1000


def test_channel_params_edges(oracle):
    N, R = 4096, 4
    # clamp at the upper band edge (py:340-341) and wrap below zero (py:338-339)
    f, l, lout, _, _ = oracle.channel_params(N, R, 0.999, 0.05)
    assert f == N - l
    f, l, _, _, _ = oracle.channel_params(N, R, 0.001, 0.05)
    assert f == (4 - l // 2 + N) % N or f == N - l
    with pytest.raises(ValueError):
        oracle.channel_params(N, R, 0.3, 0.0)   # nextpow2(0) raises (py:38-39)
    assert oracle.nextpow2(204.8) == 256 and oracle.nextpow2(256) == 256 and oracle.nextpow2(1) == 1


def test_overlap_save_blocks_and_history(oracle):
    """lib/overlap_save_impl.cc: item i = [ovl samples before it | H new]; zero history first; state kept."""
    N, ovl = 16, 4
    H = N - ovl
    x = (np.arange(5 * H) + 1).astype(np.complex64)
    blk = oracle.OverlapSave(8, N, ovl)
    a = blk.work(x[:2 * H]).reshape(2, N)
    b = blk.work(x[2 * H:]).reshape(3, N)
    xp = np.concatenate([np.zeros(ovl, np.complex64), x])
    for m, row in enumerate(np.concatenate([a, b])):
        assert (row == xp[m * H:m * H + N]).all()
    # type-agnostic: 1-byte items
    blk8 = oracle.OverlapSave(1, 8, 2)
    y = blk8.work(np.arange(12, dtype=np.uint8)).reshape(2, 8)
    assert list(y[0]) == [0, 0, 0, 1, 2, 3, 4, 5] and list(y[1]) == [4, 5, 6, 7, 8, 9, 10, 11]


def test_vector_cut(oracle):
    x = np.arange(3 * 10, dtype=np.complex64)
    y = oracle.vector_cut(8, 10, 3, 4, x).reshape(3, 4)
    assert (y == x.reshape(3, 10)[:, 3:7]).all()


def test_phase_window_counter_cycles(oracle):
    """lib/phase_shifting_windowing_vcc_impl.cc:80-83: counter += shift mod R per item."""
    l, R = 16, 4
    pw = oracle.PhaseWindow(l, R, 7, 0.5, 0.9, 1)      # shift = 7 mod 4 = 3
    x = np.ones(6 * l, np.complex64)
    y = pw.work(x).reshape(6, l)
    for m in range(6):
        assert (y[m] == pw.win[(3 * m) % R]).all()
    with pytest.raises(ValueError):
        oracle.PhaseWindow(l, R, 0, 0.9, 0.5, 1)       # stopbw < passbw (impl.cc:52-53)


def test_fft_vcc_matches_numpy(oracle):
    rng = np.random.default_rng(3)
    for n in (2, 8, 64, 512, 4096, 32768):
        x = (rng.standard_normal((3, n)) + 1j * rng.standard_normal((3, n))).astype(np.complex64)
        fwd = oracle.fft_vcc(n, True, True, x).reshape(3, n)
        ref = np.fft.fftshift(np.fft.fft(x.astype(np.complex128), axis=1), axes=1)
        assert max(rel_err(fwd, ref)) < 2e-7
        inv = oracle.fft_vcc(n, False, True, x).reshape(3, n)
        refi = np.fft.ifft(np.fft.ifftshift(x.astype(np.complex128), axes=1), axis=1) * n
        assert max(rel_err(inv, refi)) < 2e-7
        plain = oracle.fft_vcc(n, True, False, x).reshape(3, n)
        assert max(rel_err(plain, np.fft.fft(x.astype(np.complex128), axis=1))) < 2e-7


def test_chain_vs_numpy_golden(oracle, golden_dir):
    """Whole chain (C oracle) == independent numpy.fft evaluation committed in tests/golden."""
    z = np.load(os.path.join(golden_dir, "chain_numpy.npz"))
    for case in json.loads(str(z["cases"])):
        chans = [tuple(c) for c in case["chans"]]
        outs, _ = oracle.channelizer(case["N"], case["R"], case["wintype"], chans, z[case["name"] + "_x"])
        for i, o in enumerate(outs):
            ref = z[case["name"] + "_out%d" % i]
            assert o.shape == ref.shape
            l2, mx = rel_err(o, ref)
            assert l2 < 1e-6 and mx < 1e-6, (case["name"], i, l2, mx)


def test_chain_blockwise_equals_composed_blocks(oracle):
    """fdco_channelizer == overlap_save -> fft_vcc -> *1/N -> vector_cut -> phase window -> fft_vcc -> cut -> *l
    composed from the single-block restatements exactly as python/FrequencyDomainChannelizer.py:283-299 wires them."""
    N, R = 1024, 4
    H = N - N // R
    rng = np.random.default_rng(11)
    x = (rng.standard_normal(5 * H) + 1j * rng.standard_normal(5 * H)).astype(np.complex64)
    f, l, p, s = 301, 64, 0.6, 0.85
    outs, spec = oracle.channelizer(N, R, 2, [(f, l, p, s)], x, want_spectrum=True)
    blk = oracle.OverlapSave(8, N, N // R).work(x)
    S = oracle.fft_vcc(N, True, True, blk)
    X = (S * np.float32(1.0 / N)).astype(np.complex64)
    assert (X.view(np.uint32) == spec.view(np.uint32)).all()
    c = oracle.vector_cut(8, N, f, l, X)
    y = oracle.PhaseWindow(l, R, f, p, s, 2).work(c)
    t = oracle.fft_vcc(l, False, True, y)
    lout = l - l // R
    z = oracle.vector_cut(8, l, l - lout, lout, t) * np.float32(l)
    assert (z.view(np.uint32) == outs[0].view(np.uint32)).all()


def test_tone_is_continuous_across_blocks(oracle):
    """Property: an in-band bin-centred tone comes out as a unit-amplitude complex exponential,
    phase-continuous over block boundaries, for odd f and R in {2,4,8} and all window types."""
    N = 2048
    for R in (2, 4, 8):
        for wt in (0, 1, 2):
            H = N - N // R
            f, l = 777, 128
            k0 = f + l // 2 + 9
            n = np.arange(10 * H)
            x = np.exp(2j * np.pi * (k0 - N / 2) * n / N).astype(np.complex64)
            (y,), _ = oracle.channelizer(N, R, wt, [(f, l, 0.6, 0.85)], x)
            lout = l - l // R
            y = y[2 * lout:]
            assert abs(np.abs(y) - 1).max() < 1e-5
            inst = np.angle(y[1:] * np.conj(y[:-1]))
            assert abs(inst - 2 * np.pi * 9 / l).max() < 1e-5


def test_sharded_equals_whole(oracle):
    """Time-sharding property: span with halo prefix + first_block == the whole run."""
    N, R = 1024, 2
    H = N - N // R
    rng = np.random.default_rng(5)
    x = (rng.standard_normal(9 * H) + 1j * rng.standard_normal(9 * H)).astype(np.complex64)
    chans = [(33, 64, 0.6, 0.85), (512, 256, 0.88, 1.0)]
    whole, _ = oracle.channelizer(N, R, 1, chans, x)
    a, _ = oracle.channelizer(N, R, 1, chans, x[:4 * H])
    b, _ = oracle.channelizer(N, R, 1, chans, x[4 * H:], prefix=x[4 * H - N // R:4 * H], first_block=4)
    for w, p, q in zip(whole, a, b):
        assert (np.concatenate([p, q]).view(np.uint32) == w.view(np.uint32)).all()
