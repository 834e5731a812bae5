"""GPU parity tests (-m gpu): the HIP path, called through the C-ABI, against the oracle on the same
seeded inputs and against the committed golden fixtures.

Tolerance (BASELINE.json north_star / SURVEY.md §8d): relative L2 error <= 1e-5 AND
max|y - ref| / max|ref| <= 1e-5 on complex float32 streams; byte-copy blocks are bit-exact."""
import json
import os

import numpy as np
import pytest

import gr_fdc_amd as G

pytestmark = pytest.mark.gpu
TOL = 1e-5


def rel(a, b):
    a = a.astype(np.complex128); b = b.astype(np.complex128)
    nb = np.linalg.norm(b)
    if nb == 0:
        return float(np.linalg.norm(a)), float(np.abs(a).max(initial=0))
    return float(np.linalg.norm(a - b) / nb), float(np.abs(a - b).max() / np.abs(b).max())


def assert_close(a, b, what=""):
    assert a.shape == b.shape, what
    l2, mx = rel(a, b)
    assert l2 <= TOL and mx <= TOL, "%s: l2=%.3g max=%.3g" % (what, l2, mx)


def noise(n, seed):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)


# ---------------------------------------------------------------- single-block faces
def test_overlap_save_block_bit_exact(oracle):
    for itemsize, N, ovl in [(8, 64, 16), (4, 256, 128), (1, 40, 7), (8, 4096, 2048), (2, 10, 5)]:
        H = N - ovl
        raw = np.random.default_rng(N).integers(0, 256, size=7 * H * itemsize, dtype=np.uint8)
        a, b = G.overlap_save(itemsize, N, ovl), oracle.OverlapSave(itemsize, N, ovl)
        for lo, hi in [(0, 3), (3, 4), (4, 7)]:          # state carries across calls
            x = raw[lo * H * itemsize:hi * H * itemsize]
            assert (a.work(x) == b.work(x)).all()


def test_vector_cut_block_bit_exact(oracle):
    for itemsize, veclen, off, bl in [(8, 4096, 2413, 256), (8, 256, 64, 192), (1, 33, 5, 9), (4, 100, 0, 100)]:
        x = np.random.default_rng(veclen).integers(0, 256, size=5 * veclen * itemsize, dtype=np.uint8)
        got = G.vector_cut_vxx(itemsize, veclen, off, bl).work(x)
        assert (got == oracle.vector_cut(itemsize, veclen, off, bl, x)).all()


def test_phase_window_block(oracle):
    for l, R, shifts, p, s, wt in [(256, 2, 1, 0.88, 1.0, 1), (64, 4, 7, 0.5, 0.8, 2), (16, 8, -3, 0.6, 0.85, 0),
                                   (1024, 4, 2, 0.528, 0.778, 1)]:
        a, b = G.phase_shifting_windowing_vcc(l, R, shifts, p, s, wt), oracle.PhaseWindow(l, R, shifts, p, s, wt)
        for n in (3, 1, 5):
            x = noise(n * l, l + n)
            ya, yb = a.work(x), b.work(x)
            assert np.abs(ya - yb).max() <= 1e-6 * np.abs(yb).max()


@pytest.mark.parametrize("n", [2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 65536, 262144, 1048576, 4194304])
def test_fft_vcc_all_sizes(oracle, n):
    items = 3 if n <= 65536 else 2 if n <= (1 << 20) else 1
    x = noise(items * n, n)
    for fwd in (True, False):
        for shift in (True, False):
            assert_close(G.fft_vcc(n, fwd, shift, x), oracle.fft_vcc(n, fwd, shift, x), "n=%d fwd=%d shift=%d" % (n, fwd, shift))


# ---------------------------------------------------------------- fused pipeline
def test_golden_chain_fixtures(golden_dir):
    """HIP pipeline vs the committed numpy.fft goldens (cfg1 plans incl. odd f, R=2/4/8, all windows)."""
    z = np.load(os.path.join(golden_dir, "chain_numpy.npz"))
    for case in json.loads(str(z["cases"])):
        p = G.Pipeline(case["N"], case["R"], [tuple(c) for c in case["chans"]], windowtype=case["wintype"],
                       max_blocks=case["nblocks"])
        outs = p.work(z[case["name"] + "_x"])
        for i, o in enumerate(outs):
            assert_close(o, z[case["name"] + "_out%d" % i], "%s ch%d" % (case["name"], i))


@pytest.mark.parametrize("N,R,nb", [(4096, 2, 9), (4096, 4, 9), (1024, 8, 5), (64, 2, 4), (16384, 2, 5),
                                    (65536, 2, 4), (32768, 4, 3)])
def test_pipeline_vs_oracle_mixed_plan(oracle, N, R, nb):
    rng = np.random.default_rng(N + R)
    chans = []
    for _ in range(7):
        l = int(2 ** rng.integers(1, min(12, int(np.log2(N))) + 1))
        f = int(rng.integers(0, N - l + 1))
        p = float(rng.uniform(0.3, 0.95)); s = float(min(1.0, p + rng.uniform(0.05, 0.4)))
        chans.append((f, l, p, s))
    chans.append((0, 1, 0.5, 1.0))                  # degenerate 1-bin channel
    chans.append((N - 2, 2, 1.0, 1.0))              # touches the band edge, pass band 1 -> rectangular
    H = N - N // R
    x = noise(nb * H, N)
    for wt in (0, 1, 2):
        p = G.Pipeline(N, R, chans, windowtype=wt, max_blocks=nb, keep_spectrum=True)
        outs, spec = p.work(x, want_spectrum=True)
        ref, rspec = oracle.channelizer(N, R, wt, chans, x, want_spectrum=True)
        assert_close(spec, rspec, "spectrum")
        for c, (o, r) in enumerate(zip(outs, ref)):
            assert_close(o, r, "N=%d R=%d wt=%d ch%d %s" % (N, R, wt, c, chans[c]))


def test_state_carries_across_work_calls(oracle):
    """Ragged call sizes: history (overlap_save_impl.h:33) and window counter (…windowing_vcc_impl.h:47)
    persist; results equal one big call."""
    N, R = 4096, 4
    H = N - N // R
    chans = [(2413, 256, 0.88, 1.0), (963, 1024, 0.528, 0.778), (7, 64, 0.6, 0.85)]
    x = noise(11 * H, 99)
    ref, _ = oracle.channelizer(N, R, 1, chans, x)
    p = G.Pipeline(N, R, chans, windowtype=1, max_blocks=6)
    parts = [p.work(x[a * H:b * H]) for a, b in [(0, 1), (1, 6), (6, 6), (6, 8), (8, 11)]]
    for c in range(len(chans)):
        assert_close(np.concatenate([q[c] for q in parts]), ref[c], "ch%d" % c)
    p.reset()
    again = p.work(x[:3 * H])
    for c in range(len(chans)):
        assert_close(again[c], ref[c][:3 * p.lout[c]], "after reset ch%d" % c)


def test_cfg2_tiled_256_channels_vs_oracle(oracle):
    """BASELINE configs[1] at a size the oracle finishes in seconds: N=65536, R=2, 256 channels l=256."""
    N, R, Cn, nb = 65536, 2, 256, 3
    chans = [G.get_opt_channelparams(N, R, ((c + 0.5) / Cn - 0.5 + 0.5) % 1.0, 0.8 / Cn) for c in range(Cn)]
    assert all(ch[:3] == (256 * c, 256, 128) for c, ch in enumerate(chans))
    plan = [(f, l, p, s) for (f, l, _lo, p, s) in chans]
    x = noise(nb * (N - N // R), 2025)
    outs = G.Pipeline(N, R, plan, windowtype=1, max_blocks=nb).work(x)
    ref, _ = oracle.channelizer(N, R, 1, plan, x, nthreads=8)
    for c in range(Cn):
        assert_close(outs[c], ref[c], "ch%d" % c)


def test_chunking_is_invisible(oracle):
    N, R = 16384, 2
    chans = [(100, 256, 0.88, 1.0), (5001, 512, 0.7, 0.95)]
    x = noise(10 * (N - N // R), 5)
    a = G.Pipeline(N, R, chans, max_blocks=10, chunk_blocks=3).work(x)
    b = G.Pipeline(N, R, chans, max_blocks=10, chunk_blocks=10).work(x)
    for p, q in zip(a, b):
        assert (p.view(np.uint32) == q.view(np.uint32)).all()


def test_hier_block_mirror(oracle):
    """FrequencyDomainChannelizer face with the example flowgraph's parameters (examples/FDC_example.grc)."""
    user = [[0.12, 0.05], [0.22, 0.1], [-0.14, 0.12], [0, 0.081]]
    fdc = G.FrequencyDomainChannelizer(8, 1, 2 ** 12, 4, user, None, 6.0, 1.0, 0.0, 'normalized', 1,
                                       False, False, "", False, None, 10.0, 0.005, 1, 0.2, 0, 0, 128, 128, True)
    assert [cp[:3] for cp in fdc.channel_params] == [(2412, 256, 192), (2693, 512, 384), (963, 1024, 768), (1792, 512, 384)]
    x = noise(6 * fdc.inpblocklen, 42)
    ports = fdc.work(x)
    plan = [(f, l, p, s) for (f, l, _lo, p, s) in fdc.channel_params]
    ref, rspec = oracle.channelizer(4096, 4, 1, plan, x, want_spectrum=True)
    assert_close(ports[0], rspec, "debug spectrum port")
    for c in range(4):
        assert_close(ports[1 + c], ref[c], "port %d" % (1 + c))


@pytest.mark.parametrize("N,R", [(65536, 2), (65536, 4), (65536, 8), (4096, 2), (16384, 4)])
def test_fast_l256_kernels_all_alignment_variants(oracle, N, R):
    """The size-specialised kernels (N=65536 forward, l=256 channels): odd/even slice offsets, odd/even
    output offsets (an l=1 channel in front makes every later offset odd), phase rotation for odd f."""
    nb = 5
    for lead in ([], [(0, 1, 0.5, 1.0)]):
        chans = lead + [(3, 256, 0.88, 1.0), (512, 256, 0.6, 0.9), (1025, 256, 0.88, 1.0), (N - 256, 256, 0.7, 0.95),
                        (2050, 256, 1.0, 1.0), (777, 128, 0.8, 1.0)]
        x = noise(nb * (N - N // R), N + R + len(lead))
        for wt in (1, 2):
            outs = G.Pipeline(N, R, chans, windowtype=wt, max_blocks=nb).work(x)
            ref, _ = oracle.channelizer(N, R, wt, chans, x, nthreads=4)
            for c, (o, r) in enumerate(zip(outs, ref)):
                assert_close(o, r, "N=%d R=%d lead=%d wt=%d ch%d" % (N, R, len(lead), wt, c))


def test_fast_and_generic_paths_agree(oracle):
    """FDC_FORCE_GENERIC=1 routes the same call through the generic LDS Stockham kernels."""
    N, R, nb = 65536, 2, 3
    chans = [(256 * c, 256, 0.88, 1.0) for c in range(0, 256, 17)] + [(1, 256, 0.88, 1.0)]
    x = noise(nb * (N - N // R), 77)
    fast = G.Pipeline(N, R, chans, max_blocks=nb, keep_spectrum=True).work(x, want_spectrum=True)
    G.defaults["FDC_FORCE_GENERIC"] = "1"
    try:
        slow = G.Pipeline(N, R, chans, max_blocks=nb, keep_spectrum=True).work(x, want_spectrum=True)
    finally:
        del G.defaults["FDC_FORCE_GENERIC"]
    assert_close(fast[1], slow[1], "spectrum")
    for a, b in zip(fast[0], slow[0]):
        assert_close(a, b)


@pytest.mark.parametrize("N,R,wt", [(65536, 2, 1), (65536, 4, 2), (65536, 8, 0), (65536, 16, 1),
                                    (262144, 2, 1), (262144, 4, 0), (262144, 16, 2),
                                    (4096, 2, 1), (8192, 4, 2), (16384, 2, 0), (32768, 16, 1), (131072, 2, 1),
                                    (524288, 8, 2), (1048576, 2, 1)])
def test_uniform_plan_two_stage_path(oracle, N, R, wt):
    """Uniform plans (l=256 on the 256-bin grid; 256 slots at N=65536, 1024 at N=262144) take the two-stage path that
    never writes a spectrum; any subset and any order of slots; result equals the oracle and the spectrum-in-memory path."""
    nb = 5
    slots = {65536: [200, 3, 255, 0, 17, 128, 127, 64], 262144: [200, 3, 1023, 0, 517, 512, 511, 64, 900, 256, 767]}.get(N)
    if slots is None:                                  # other slot counts: stage 2 on the generic core
        n1 = N // 256
        slots = [int(v) for v in np.random.default_rng(N).permutation(n1)[:min(n1, 9)]] + [0, n1 - 1, n1 // 2]
        slots = list(dict.fromkeys(slots))
    chans = [(256 * c, 256, 0.88, 1.0) for c in slots]
    x = noise(nb * (N - N // R), 31 + R)
    p = G.Pipeline(N, R, chans, windowtype=wt, max_blocks=nb, chunk_blocks=2)
    forced = any(G.defaults.get(k) for k in ("FDC_FORCE_GENERIC", "FDC_NO_POLY"))   # the suite run under a forced path
    if N == 4096 and not forced and not G.defaults.get("FDC_NO_FUSED"):
        # round 6: N = 4096 runs in ONE launch with the spectrum in LDS (path 5, tests/test_fused4096_gpu.py); the two-stage form is what FDC_PIPE_NO_FUSED leaves
        assert p.path() == 5
        p = G.Pipeline(N, R, chans, windowtype=wt, max_blocks=nb, chunk_blocks=2, flags=G.FDC_PIPE_NO_FUSED)
    assert forced or p.path() == (3 if N in (16384, 32768, 65536) and R in (2, 4) and not G.defaults.get("FDC_NO_BLOCK") else 2)   # N = 16384 / 32768 / 65536, R = 2 or 4: the one-kernel form
    outs = p.work(x)
    ref, _ = oracle.channelizer(N, R, wt, chans, x, nthreads=4)
    for c in range(len(chans)):
        assert_close(outs[c], ref[c], "slot %d" % slots[c])
    G.defaults["FDC_NO_POLY"] = "1"
    try:
        q = G.Pipeline(N, R, chans, windowtype=wt, max_blocks=nb)
        # (round 5: the block kernel is the forward transform at all three; FDC_NO_BLOCK leaves the register kernels of N = 65536)
        assert forced or q.path() == (1 if N == 65536 or (N in (16384, 32768) and not G.defaults.get("FDC_NO_BLOCK")) else 0)
        outs3 = q.work(x)
    finally:
        del G.defaults["FDC_NO_POLY"]
    for a, b in zip(outs, outs3):
        assert_close(a, b)
    # first_block / history handling on the two-stage path: ragged calls equal one call
    p.reset()
    parts = [p.work(x[a * p.H:b * p.H]) for a, b in [(0, 2), (2, 3), (3, 5)]]
    for c in range(len(chans)):
        assert_close(np.concatenate([q_[c] for q_ in parts]), ref[c])


@pytest.mark.parametrize("nslots,nb,chunk", [(256, 7, 0), (256, 300, 0), (256, 530, 256), (9, 261, 0), (1, 3, 0)])
def test_uniform_plan_one_kernel_path(oracle, nslots, nb, chunk):
    """N = 65536, R = 2, uniform plan: the one-block-per-CU kernel (path 3; G never leaves the compute unit) against the
    oracle on the first and last blocks and against the two-launch path (FDC_NO_BLOCK=1) on every sample; block counts
    below, at and above one round of workgroups, launch groups that end mid-round, any subset of slots."""
    N, R = 65536, 2
    H = N - N // R
    rng = np.random.default_rng(nslots * 1000 + nb)
    slots = [int(v) for v in rng.permutation(256)[:nslots]]
    chans = [(256 * c, 256, 0.88, 1.0) for c in slots]
    x = noise(nb * H, 77 + nb)
    G.defaults["FDC_HOST_SUB"] = str(nb)              # the whole call as one device batch (launch groups of `chunk` blocks)
    p = G.Pipeline(N, R, chans, windowtype=1, max_blocks=nb, chunk_blocks=chunk)
    if any(G.defaults.get(k) for k in ("FDC_FORCE_GENERIC", "FDC_NO_POLY", "FDC_NO_BLOCK")):
        pytest.skip("suite run under a forced path")
    assert p.path() == 3
    try:
        outs = p.work(x)
    finally:
        G.defaults.pop("FDC_HOST_SUB", None)
    k = min(nb, 3)
    ref, _ = oracle.channelizer(N, R, 1, chans, x[:k * H], nthreads=8)
    for c in range(len(chans)):
        assert_close(outs[c][:k * 128], ref[c], "slot %d head" % slots[c])
    if nb > k:                                        # last blocks, with the true history in front of them
        t0 = nb - k
        ref2, _ = oracle.channelizer(N, R, 1, chans, x[t0 * H:], prefix=x[t0 * H - N // R:t0 * H], first_block=t0, nthreads=8)
        for c in range(len(chans)):
            assert_close(outs[c][t0 * 128:], ref2[c], "slot %d tail" % slots[c])
    G.defaults["FDC_NO_BLOCK"] = "1"
    try:
        q = G.Pipeline(N, R, chans, windowtype=1, max_blocks=nb, chunk_blocks=chunk)
        assert q.path() == 2
        outs2 = q.work(x)
    finally:
        del G.defaults["FDC_NO_BLOCK"]
    for c in range(len(chans)):
        assert_close(outs[c], outs2[c], "slot %d vs two-launch path" % slots[c])


@pytest.mark.parametrize("r,nslots", [(1, 255), (16, 40), (37, 255), (128, 7), (255, 255), (200, 1)])
def test_offset_uniform_plan_one_kernel_path(oracle, r, nslots):
    """A tiling that does not start at bin 0 — every channel at f = 256*slot + r, one window — runs on the one-kernel path too
    (the block modulated by exp(-2 pi i r n / N) inside the kernel; for odd r the window phase alternates per block,
    lib/phase_shifting_windowing_vcc_impl.cc:57-58,82).  Against the oracle, against the spectrum-in-memory path, and in
    ragged calls (the phase depends on the GLOBAL block index)."""
    N, R, nb = 65536, 2, 7
    H = N - N // R
    rng = np.random.default_rng(r * 31 + nslots)
    slots = [int(v) for v in rng.permutation(255)[:nslots]]          # slot 255 would leave the spectrum for r > 0
    chans = [(256 * c + r, 256, 0.88, 1.0) for c in slots]
    x = noise(nb * H, 11 + r)
    if any(G.defaults.get(k) for k in ("FDC_FORCE_GENERIC", "FDC_NO_POLY", "FDC_NO_BLOCK")):
        pytest.skip("suite run under a forced path")
    p = G.Pipeline(N, R, chans, windowtype=1, max_blocks=nb)
    assert p.path() == 3
    outs = p.work(x)
    ref, _ = oracle.channelizer(N, R, 1, chans, x, nthreads=8)
    for c in range(0, len(chans), max(1, len(chans) // 16)):
        assert_close(outs[c], ref[c], "slot %d offset %d" % (slots[c], r))
    G.defaults["FDC_NO_POLY"] = "1"
    try:
        q = G.Pipeline(N, R, chans, windowtype=1, max_blocks=nb)
        assert q.path() == 1
        outs3 = q.work(x)
    finally:
        del G.defaults["FDC_NO_POLY"]
    for a, b in zip(outs, outs3):
        assert_close(a, b)
    p.reset()
    parts = [p.work(x[a * H:b * H]) for a, b in [(0, 1), (1, 4), (4, 7)]]     # calls starting at odd and even block indices
    for c in range(len(chans)):
        assert_close(np.concatenate([q_[c] for q_ in parts]), outs[c])
    # without the one-kernel form an offset plan is not a uniform plan
    G.defaults["FDC_NO_BLOCK"] = "1"
    try:
        assert G.Pipeline(N, R, chans, windowtype=1, max_blocks=nb).path() == 1
    finally:
        del G.defaults["FDC_NO_BLOCK"]


def test_cfg4_262144_tiled_1024_channels_sharded_spans(oracle):
    """BASELINE configs[3] at a size the oracle finishes in seconds: N=262144, R=2, 1024 channels (l=256), processed as
    two independent block spans (halo + global first-block index) exactly as the 8-GPU sharding does, through the
    device-resident entry; spans concatenated == oracle on the whole stream."""
    import ctypes as C
    from gr_fdc_amd import _lib
    N, R, Cn, nb = 262144, 2, 1024, 3
    H = N - N // R
    params = [G.get_opt_channelparams(N, R, ((c + 0.5) / Cn) % 1.0, 0.8 / Cn) for c in range(Cn)]
    assert all(p[:3] == (256 * c, 256, 128) for c, p in enumerate(params))
    plan = [(f, l, p, s) for (f, l, _lo, p, s) in params]
    x = noise(nb * H, 2027)
    ref, _ = oracle.channelizer(N, R, 1, plan, x, nthreads=8)
    pipe = G.Pipeline(N, R, plan, windowtype=1, max_blocks=nb)
    assert pipe.path() == 2 or any(G.defaults.get(k) for k in ("FDC_FORCE_GENERIC", "FDC_NO_POLY"))   # two-stage path, 1024 slots
    whole = pipe.work(x)
    for c in range(0, Cn, 37):
        assert_close(whole[c], ref[c], "whole ch%d" % c)
    # two spans as two "ranks" would run them (stateless device entry needs raw device buffers: use hip through torch-free ctypes)
    hip = C.CDLL("libamdhip64.so")
    for first, n in [G.span_for_rank(nb, r, 2) for r in range(2)]:
        ring = np.ascontiguousarray(G.ring_for_span(x, first, n, N, R))
        d_ring, d_out = C.c_void_p(), C.c_void_p()
        nout = pipe.output_samples(n)
        assert hip.hipMalloc(C.byref(d_ring), C.c_size_t(ring.nbytes)) == 0
        assert hip.hipMalloc(C.byref(d_out), C.c_size_t(nout * 8)) == 0
        assert hip.hipMemcpy(d_ring, C.c_void_p(ring.ctypes.data), C.c_size_t(ring.nbytes), 1) == 0
        pipe.process_device(d_ring, first, n, d_out)
        pipe.synchronize()
        out = np.empty(nout, np.complex64)
        assert hip.hipMemcpy(C.c_void_p(out.ctypes.data), d_out, C.c_size_t(out.nbytes), 2) == 0
        hip.hipFree(d_ring); hip.hipFree(d_out)
        for c in range(0, Cn, 41):
            o = out[pipe.channel_offset(c, n):pipe.channel_offset(c, n) + n * 128]
            assert_close(o, ref[c][first * 128:(first + n) * 128], "span first=%d ch%d" % (first, c))


def test_hier_block_inpveclen_mode(oracle):
    """inpveclen = blocksize: the caller did overlap-save + forward FFT (python/FrequencyDomainChannelizer.py:284-290)."""
    N, R, nb = 4096, 4, 5
    H = N - N // R
    x = noise(nb * H, 71)
    user = [[0.12, 0.05], [-0.14, 0.12]]
    items = oracle.fft_vcc(N, True, True, oracle.OverlapSave(8, N, N // R).work(x))      # what the caller's front end delivers
    fdc = G.FrequencyDomainChannelizer(8, N, N, R, user, None, 6.0, 1.0, 0.0, 'normalized', 2, False, False, "", False,
                                       None, 10.0, 0.005, 1, 0.2, 0, 0, 128, 128, True, max_blocks=nb)
    ports = fdc.work(items)
    plan = [(f, l, p, s) for (f, l, _lo, p, s) in fdc.channel_params]
    ref, rspec = oracle.channelizer(N, R, 2, plan, x, want_spectrum=True)
    assert_close(ports[0], rspec, "normalised spectrum port")
    for c in range(2):
        assert_close(ports[1 + c], ref[c], "port %d" % (1 + c))


@pytest.mark.parametrize("sub", ["1", "2", "64"])
def test_host_path_sub_batches_staged_and_pinned(oracle, sub):
    """fdc_pipeline_work cuts a call into sub-batches whose transfers and kernels overlap; pageable buffers are staged,
    buffers pinned with fdc_host_register are DMA'd in place (outputs stored by a scatter kernel).  Every combination
    must give the oracle's samples, across calls (history + block counter carry over)."""
    N, R, nb = 4096, 4, 7
    H = N - N // R
    chans = [(8, 512, 0.8, 1.0), (1031, 64, 0.7, 0.9), (3000, 1024, 0.5, 0.75), (777, 1, 0.5, 1.0), (2048, 256, 0.9, 1.0)]
    x = noise(2 * nb * H, 77)
    ref, _ = oracle.channelizer(N, R, 1, chans, x, nthreads=2)
    G.defaults["FDC_HOST_SUB"] = sub
    try:
        p = G.Pipeline(N, R, chans, windowtype=1, max_blocks=nb)
        los = p.lout
        # call 1 staged (pageable numpy), call 2 pinned input + pinned outputs
        o1 = p.work(x[:nb * H])
        xin = np.empty(nb * H, np.complex64); xin[:] = x[nb * H:]
        pool = np.empty(nb * sum(los), np.complex64)
        G.register_host(xin); G.register_host(pool)
        try:
            outs, off = [], 0
            for lo in los:
                outs.append(pool[off:off + nb * lo]); off += nb * lo
            o2 = p.work(xin, outs=outs)
            o2 = [o.copy() for o in o2]
            # pinned input, pageable outputs and the other way round
            p.reset()
            o3 = p.work(x[:nb * H], outs=outs); o3 = [o.copy() for o in o3]
            o4 = p.work(xin)
        finally:
            G.unregister_host(xin); G.unregister_host(pool)
    finally:
        del G.defaults["FDC_HOST_SUB"]
    for c, lo in enumerate(los):
        assert_close(np.concatenate([o1[c], o2[c]]), ref[c], "ch%d" % c)
        assert_close(np.concatenate([o3[c], o4[c]]), ref[c], "ch%d (mixed)" % c)


def test_widest_channels_and_no_channels(oracle):
    """Every channel width up to the whole band: above 4096 bins the channels of one width and all blocks of a launch group are one
    batch of the task-addressed two-pass inverse transform; a plan without any channel (spectrum only); an empty call."""
    N, R, nb = 32768, 4, 3
    H = N - N // R
    chans = [(100, 8192, 0.7, 0.9), (12000, 4096, 0.5, 0.8), (20000, 2048, 0.9, 1.0), (30000, 8, 0.5, 1.0),
             (16001, 16384, 0.6, 0.8), (0, 32768, 0.9, 1.0)]
    x = noise(nb * H, 5)
    ref, rspec = oracle.channelizer(N, R, 2, chans, x, want_spectrum=True, nthreads=2)
    p = G.Pipeline(N, R, chans, windowtype=2, max_blocks=nb, keep_spectrum=True)
    outs, spec = p.work(x, want_spectrum=True)
    assert_close(spec, rspec, "spectrum")
    for c in range(len(chans)):
        assert_close(outs[c], ref[c], "l=%d" % chans[c][1])
    assert [o.size for o in p.work(x[:0])] == [0] * len(chans)
    q = G.Pipeline(N, R, [], windowtype=2, max_blocks=nb, keep_spectrum=True)
    o2, s2 = q.work(x, want_spectrum=True)
    assert o2 == [] and np.array_equal(s2, spec)
    with pytest.raises(G.FdcError):
        p.work(np.zeros((nb + 1) * H, np.complex64))          # above max_blocks
    with pytest.raises(ValueError):
        p.work(np.zeros(H + 1, np.complex64))                 # not a whole number of items
    with pytest.raises(ValueError):
        G.Pipeline(N, R, [(0, 3, 0.5, 1.0)], max_blocks=1)   # l not a power of two


def test_wide_channels_in_pieces(oracle):
    """Channels above 4096 bins, several of one width, over more blocks than one 32 Mi-point piece of the scratch between the two
    passes holds (the 32768-wide channel: 1030 blocks = two pieces); every sample of every channel against the oracle."""
    if G.defaults.get("FDC_FORCE_GENERIC"):
        pytest.skip("suite run under a forced path")
    N, R, nb = 65536, 2, 1030
    chans = [(0, 32768, 0.8, 0.95), (32768, 8192, 0.88, 1.0), (41060, 8192, 0.7, 0.9), (49152, 16384, 0.88, 1.0), (5000, 1024, 0.88, 1.0)]
    x = noise(nb * (N - N // R), 123)
    ref, _ = oracle.channelizer(N, R, 1, chans, x, nthreads=8)
    out = G.Pipeline(N, R, chans, windowtype=1, max_blocks=nb).work(x)
    for c in range(len(chans)):
        assert_close(out[c], ref[c], "l=%d" % chans[c][1])


def test_fft_vcc_largest_supported_size(oracle):
    """N = 2^24, the largest block length fdc_pipeline_create accepts: forward, shifted."""
    n = 1 << 24
    x = noise(n, 24)
    assert_close(G.fft_vcc(n, True, True, x), oracle.fft_vcc(n, True, True, x), "n=2^24")


def test_handles_release_their_device_memory(oracle):
    """create / work / destroy in a loop: device memory in use returns to where it started (every path's scratch, tables,
    streams, events, pinned staging and registrations are released)."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")

    def free_bytes():
        f, t = C.c_size_t(), C.c_size_t()
        assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value

    N, R, nb = 65536, 2, 8
    H = N - N // R
    uni = [(256 * c, 256, 0.88, 1.0) for c in range(0, 256, 3)]
    mixed = [(100, 512, 0.7, 0.9), (5000, 16384, 0.6, 0.8), (40000, 256, 0.88, 1.0)]
    x = noise(nb * H, 3)
    pool = np.empty(nb * 128 * len(uni), np.complex64)
    G.register_host(pool)
    outs = [pool[i * nb * 128:(i + 1) * nb * 128] for i in range(len(uni))]
    before = None
    for it in range(6):
        p = G.Pipeline(N, R, uni, max_blocks=nb); p.work(x, outs=outs); p.enable_timing(True); p.work(x); p.last_kernel_ms(); p.close()
        q = G.Pipeline(N, R, mixed, max_blocks=nb, keep_spectrum=True)
        s = G.Sinks(N, R, pac=[(0.3, 0.01, 1)], pac_thresh=6.0, pac_maxblocks=-1, segments=[(0.5, 0.9)], det_thresh=10.0,
                    det_maxblocks=-1, minchandist=0.005, det_delay=1, puffer=0.2, max_blocks=nb)
        q.work(x, want_spectrum=True, sinks=s); s.close(); q.close()
        if it == 1:
            before = free_bytes()            # after the runtime's own pools have warmed up
    G.unregister_host(pool)
    assert before is not None and abs(free_bytes() - before) <= (64 << 20)


def test_distinct_handles_from_concurrent_threads(oracle):
    """One handle per thread, used concurrently (GNU Radio's thread-per-block model): results equal the single-threaded ones."""
    import threading
    N, R, nb = 16384, 4, 6
    H = N - N // R
    plans = [[(256 * c, 256, 0.88, 1.0) for c in range(64)],                        # uniform path (64 slots)
             [(100, 512, 0.7, 0.9), (5000, 2048, 0.6, 0.8), (12000, 64, 0.5, 1.0)],  # generic path
             [(256 * c + 3, 256, 0.8, 1.0) for c in range(0, 60, 7)],                # off-grid l = 256
             [(0, 16384, 0.9, 1.0)]]                                                 # one channel as wide as the band
    xs = [noise(3 * nb * H, 50 + i) for i in range(len(plans))]
    want = []
    for pl, x in zip(plans, xs):
        p = G.Pipeline(N, R, pl, max_blocks=nb)
        want.append([np.concatenate(parts) for parts in zip(*[p.work(x[k * nb * H:(k + 1) * nb * H]) for k in range(3)])])
        p.close()
    got = [None] * len(plans)

    def worker(i):
        p = G.Pipeline(N, R, plans[i], max_blocks=nb)
        got[i] = [np.concatenate(parts) for parts in zip(*[p.work(xs[i][k * nb * H:(k + 1) * nb * H]) for k in range(3)])]
        p.close()

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(len(plans))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for i in range(len(plans)):
        assert got[i] is not None
        for a, b in zip(got[i], want[i]):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "plan %d differs under concurrency" % i


@pytest.mark.parametrize("N,R", [(4096, 4), (65536, 2)])
def test_real_input_front_end(oracle, N, R):
    """f4: float32 input items (the hier block's Float input type, python/FrequencyDomainChannelizer.py:207-208): the chain on
    x + 0j.  State carries across calls; through the hier-block mirror with inptype = 4 as well."""
    H = N - N // R
    nb = 5
    chans = [(256 * c, 256, 0.88, 1.0) for c in (3, 9, N // 256 - 1)] if N == 65536 else [(100, 256, 0.88, 1.0), (2001, 512, 0.7, 0.95)]
    rng = np.random.default_rng(N)
    x = rng.standard_normal(nb * H).astype(np.float32)
    ref, rspec = oracle.channelizer(N, R, 1, chans, x.astype(np.complex64), want_spectrum=True, nthreads=4)
    p = G.Pipeline(N, R, chans, windowtype=1, max_blocks=nb, keep_spectrum=True)
    a, spec = p.work_real(x[:2 * H], want_spectrum=True)
    b = p.work_real(x[2 * H:])
    for c in range(len(chans)):
        assert_close(np.concatenate([a[c], b[c]]), ref[c], "ch%d" % c)
    assert_close(spec, rspec[:2 * N])
    # a real signal has a Hermitian spectrum: bin k and bin N - k of the UNSHIFTED transform are conjugates (shifted: N/2 +- d)
    s0 = spec[:N]
    assert np.abs(s0[N // 2 + 5] - np.conj(s0[N // 2 - 5])) <= 1e-5 * np.abs(s0).max()
    q = G.Pipeline(N, R, chans, windowtype=1, max_blocks=nb)            # without a spectrum consumer: the fast paths
    c2 = q.work_real(x)
    for c in range(len(chans)):
        assert_close(c2[c], ref[c], "fast path ch%d" % c)


@pytest.mark.parametrize("N,R,nb", [(65536, 2, 5), (65536, 4, 5), (32768, 2, 5), (32768, 4, 261), (16384, 2, 7), (16384, 4, 530)])
def test_spectrum_path_block_forward_kernel_vs_two_pass(oracle, N, R, nb):
    """N = 65536 (round 5: and 32768 / 16384: k_blk256<P, ..., FWD>) with a mixed channel plan (spectrum in memory): the forward transform runs
    on the block kernel (both halves of k2 in one launch, fdc_block256.hip FWD); the two-pass kernels (FDC_NO_BLOCK=1) must give the same
    spectrum and the same channel outputs, and both must match the oracle.  R = 4: nothing in the forward kernel depends on R.  Block counts
    above one round of the persistent workgroups."""
    if any(G.defaults.get(k) for k in ("FDC_FORCE_GENERIC", "FDC_NO_BLOCK")):
        pytest.skip("suite run under a forced path")
    chans = [(37, 256, 0.88, 1.0), (300, 512, 0.9, 1.0), (4096, 1024, 0.88, 1.0), (N - 256, 256, 0.8, 0.95), (N // 2 - 64, 128, 0.88, 1.0)]
    x = noise(nb * (N - N // R), 77 + R)
    ref, sref = oracle.channelizer(N, R, 1, chans, x, want_spectrum=True, nthreads=4)
    res = {}
    for force in (None, "FDC_NO_BLOCK"):
        if force:
            G.defaults[force] = "1"
        try:
            p = G.Pipeline(N, R, chans, windowtype=1, max_blocks=nb, keep_spectrum=True)
            assert p.path() == (1 if (force is None or N == 65536) else 0)
            res[force] = p.work(x, want_spectrum=True)
        finally:
            if force:
                del G.defaults[force]
    for force, (outs, spec) in res.items():
        assert_close(spec.reshape(-1), sref.reshape(-1), "spectrum (%s)" % force)
        for c in range(len(chans)):
            assert_close(outs[c], ref[c], "channel %d (%s)" % (c, force))
    assert_close(res[None][1].reshape(-1), res["FDC_NO_BLOCK"][1].reshape(-1), "block kernel vs two-pass spectrum")


@pytest.mark.parametrize("N", [4096, 65536, 32768, 16384])
def test_plans_that_read_part_of_the_band_leave_the_rest_unwritten(oracle, N):
    """A few channels in a wide band: the forward kernels whose waves store whole 64-bin runs (k_fft4096, the block kernel as a
    forward transform) write only the groups some channel reads into the handle's internal spectrum (FDC_PIPE_FULL_SPECTRUM switches
    that off).  Channels that start / end inside a 64-bin group, at bin 0 and at the top of the band, wider than a slot; outputs
    against the oracle and bit for bit against the full-spectrum form; a spectrum asked for by the caller is still complete."""
    if any(G.defaults.get(k) for k in ("FDC_FORCE_GENERIC", "FDC_NO_BLOCK")):
        pytest.skip("suite run under a forced path")
    R, nb = 2, 6
    if N == 4096:
        chans = [(0, 64, 0.88, 1.0), (100, 256, 0.9, 1.0), (1023, 128, 0.88, 1.0), (2048 + 63, 512, 0.8, 0.95), (4096 - 32, 32, 0.88, 1.0)]
    else:
        chans = [(0, 256, 0.88, 1.0), (37, 256, 0.88, 1.0), (300, 512, 0.9, 1.0), (N // 4 + 191, 2048, 0.88, 1.0), (5 * N // 8 + 17, 64, 0.8, 0.95),
                 (N - 1024, 1024, 0.88, 1.0)]
    x = noise(nb * (N - N // R), 91)
    ref, sref = oracle.channelizer(N, R, 1, chans, x, want_spectrum=True, nthreads=4)
    # (round 6: this plan at N = 4096 runs without a spectrum in memory by default; the test is about the spectrum path's partial writes)
    nf = G.FDC_PIPE_NO_FUSED if N == 4096 else 0
    part = G.Pipeline(N, R, chans, windowtype=1, max_blocks=nb, flags=nf).work(x)
    full = G.Pipeline(N, R, chans, windowtype=1, max_blocks=nb, flags=G.FDC_PIPE_FULL_SPECTRUM | nf).work(x)
    for c in range(len(chans)):
        assert_close(part[c], ref[c], "channel %d" % c)
        assert (part[c] == full[c]).all(), c
    outs, spec = G.Pipeline(N, R, chans, windowtype=1, max_blocks=nb, keep_spectrum=True).work(x, want_spectrum=True)
    assert_close(spec.reshape(-1), sref.reshape(-1), "spectrum handed to the caller")
    for c in range(len(chans)):
        assert (outs[c] == full[c]).all(), c


def test_plan_classes_one_kernel_path(oracle):
    """A plan that is the union of a few 256-bin tilings stays on the one-kernel path, one launch per class (fdc_api.hip,
    PolyClass): (a) a 2x oversampled bank (tilings at offsets 0 and 128, 200 + 200 channels, interleaved in plan order),
    (b) the same slots with two different windows, (c) the same slot twice with the same window (two classes).  Plans whose
    classes would cost more than the spectrum path (few channels in several classes, or more than three classes) fall back.
    Every output is checked against the oracle and against the spectrum path."""
    if any(G.defaults.get(k) for k in ("FDC_FORCE_GENERIC", "FDC_NO_POLY", "FDC_NO_BLOCK")):
        pytest.skip("suite run under a forced path")
    N, R, nb = 65536, 2, 5
    H = N - N // R
    x = noise(nb * H, 4242)
    rng = np.random.default_rng(5)
    s0 = [int(v) for v in rng.permutation(255)[:200]]
    s1 = [int(v) for v in rng.permutation(255)[:200]]
    plan_a = [ch for pair in zip([(256 * c, 256, 0.88, 1.0) for c in s0], [(256 * c + 128, 256, 0.88, 1.0) for c in s1]) for ch in pair]
    win = [(0.7, 0.9), (0.88, 1.0)]
    plan_b = [(256 * c + 37, 256) + win[i % 2] for i, c in enumerate(s0)] + [(256 * c + 37, 256) + win[(i + 1) % 2] for i, c in enumerate(s0)]
    plan_c = [(256 * c, 256, 0.88, 1.0) for c in s0] + [(256 * c, 256, 0.88, 1.0) for c in s0[:150]]
    for name, plan in (("oversampled", plan_a), ("two windows", plan_b), ("slots twice", plan_c)):
        p = G.Pipeline(N, R, plan, windowtype=1, max_blocks=nb)
        assert p.path() == 3, name
        outs = p.work(x)
        ref, _ = oracle.channelizer(N, R, 1, plan, x, nthreads=8)
        for c in list(range(0, len(plan), 23)) + [len(plan) - 1]:
            assert_close(outs[c], ref[c], "%s: channel %d" % (name, c))
        G.defaults["FDC_NO_POLY"] = "1"
        try:
            outs3 = G.Pipeline(N, R, plan, windowtype=1, max_blocks=nb).work(x)
        finally:
            del G.defaults["FDC_NO_POLY"]
        for a, b in zip(outs, outs3):
            assert_close(a, b, name)
        p.reset()                                       # ragged calls: the odd offset's sign follows the global block index
        parts = [p.work(x[a * H:b * H]) for a, b in [(0, 2), (2, 5)]]
        for c in range(0, len(plan), 41):
            assert_close(np.concatenate([q_[c] for q_ in parts]), outs[c], name)
    # too few channels for two launches to pay: spectrum path
    assert G.Pipeline(N, R, plan_a[:40], windowtype=1, max_blocks=nb).path() == 1
    # four tilings of 200 channels: four launches still beat 3.1 bands' worth of channel kernels (round 5: up to four banks, fdc_plan_cost.hpp);
    # six: the smallest ones go back to a remainder (a split plan) — results against the spectrum path either way
    for nt in (4, 6):
        many = [(256 * c + r, 256, 0.88, 1.0) for r in (0, 64, 128, 192, 32, 96)[:nt] for c in range(200 if r in (0, 64, 128, 192) else 20)]
        p = G.Pipeline(N, R, many, windowtype=1, max_blocks=nb)
        assert p.path() == 3 if nt == 4 else p.path() in (1, 4), p.describe()      # six: whatever the cost rule makes of it
        outs = p.work(x)
        G.defaults["FDC_NO_POLY"] = "1"
        try:
            outs3 = G.Pipeline(N, R, many, windowtype=1, max_blocks=nb).work(x)
        finally:
            del G.defaults["FDC_NO_POLY"]
        for a, b in zip(outs, outs3):
            assert_close(a, b, "%d tilings" % nt)


def test_short_calls_take_the_tiled_kernels(oracle):
    """Default dispatch (no FDC_BLOCK_MIN_BLOCKS): a launch group of fewer than 96 blocks runs on the tiled kernels (two-launch
    uniform path, two-pass forward transform), 96 and more on the block kernels; a call of 100 blocks cut into groups of
    96 + 4 uses both.  All of them must agree with the oracle — and the path id stays what the plan qualifies for."""
    if any(G.defaults.get(k) for k in ("FDC_FORCE_GENERIC", "FDC_NO_POLY", "FDC_NO_BLOCK")):
        pytest.skip("suite run under a forced path")
    N, R, nb = 65536, 2, 100
    H = N - N // R
    chans = [(256 * c, 256, 0.88, 1.0) for c in (0, 5, 128, 200, 255)]
    mixed = [(37, 256, 0.88, 1.0), (4096, 512, 0.9, 1.0)]
    x = noise(nb * H, 99)
    ref, _ = oracle.channelizer(N, R, 1, chans, x, nthreads=8)
    refm, _ = oracle.channelizer(N, R, 1, mixed, x, nthreads=8)
    saved = G.defaults.pop("FDC_BLOCK_MIN_BLOCKS", None)
    try:
        for chunk in (0, 96):
            p = G.Pipeline(N, R, chans, windowtype=1, max_blocks=nb, chunk_blocks=chunk)
            assert p.path() == 3
            for calls in ([(0, 100)], [(0, 7), (7, 100)]):           # 7 blocks: tiled kernels; 93: tiled; 100: block kernel (+4 tiled)
                p.reset()
                parts = [p.work(x[a * H:b * H]) for a, b in calls]
                for c in range(len(chans)):
                    assert_close(np.concatenate([q_[c] for q_ in parts]), ref[c], "chunk %d calls %s channel %d" % (chunk, calls, c))
            q = G.Pipeline(N, R, mixed, windowtype=1, max_blocks=nb, chunk_blocks=chunk)
            assert q.path() == 1
            parts = [q.work(x[a * H:b * H]) for a, b in [(0, 3), (3, 100)]]
            for c in range(len(mixed)):
                assert_close(np.concatenate([q_[c] for q_ in parts]), refm[c], "mixed plan channel %d" % c)
    finally:
        if saved is not None:
            G.defaults["FDC_BLOCK_MIN_BLOCKS"] = saved


def test_every_offset_of_the_one_kernel_path(oracle):
    """All 255 non-zero offsets r (f = 256*slot + r): the rotation of the exchange slots by r mod 16, the second twiddle row
    table, the cbt table with (b + r) and the per-block sign of odd r, each against the oracle on three blocks."""
    if any(G.defaults.get(k) for k in ("FDC_FORCE_GENERIC", "FDC_NO_POLY", "FDC_NO_BLOCK")):
        pytest.skip("suite run under a forced path")
    N, R, nb = 65536, 2, 3
    x = noise(nb * (N - N // R), 2718)
    rng = np.random.default_rng(31)
    for r in range(1, 256):
        slots = [int(v) for v in rng.permutation(255)[:2]]
        chans = [(256 * c + r, 256, 0.88, 1.0) for c in slots]
        p = G.Pipeline(N, R, chans, windowtype=1, max_blocks=nb)
        assert p.path() == 3
        outs = p.work(x)
        ref, _ = oracle.channelizer(N, R, 1, chans, x, nthreads=4)
        for c in range(2):
            assert_close(outs[c], ref[c], "offset %d slot %d" % (r, slots[c]))


def test_randomized_plans_and_call_patterns():
    """tools/fuzz_paths.py, 24 seeded cases: random plans (on-grid, offset, classes, mixed widths, few channels), block counts
    on both sides of the dispatch threshold, chunk sizes and ragged call patterns; the default dispatch against the
    spectrum-in-memory path on every sample, mixed plans and every fifth case against the oracle."""
    import subprocess
    import sys
    if any(G.defaults.get(k) for k in ("FDC_FORCE_GENERIC", "FDC_NO_POLY", "FDC_NO_BLOCK")):
        pytest.skip("suite run under a forced path")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_paths.py"), "24", "7"], capture_output=True, text=True,
                         timeout=600, cwd=root)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-1000:]
    assert "all 24 cases within" in out.stdout


@pytest.mark.parametrize("wt", [0, 1, 2])
def test_one_kernel_form_at_relinvovl_4(oracle, wt):
    """R = 4 (the reference's default overlap, grc/FDC_FrequencyDomainChannelizer.xml:61) on the one-kernel path: 192 of the 256
    rows of every inverse transform are kept — 128 in registers, 64 through the per-workgroup scratch and a third run of
    stage 2.  All 256 slots and a scattered subset, against the oracle and against the two-launch form on every sample;
    several workgroup rounds (more blocks than a small grid would take at once), ragged calls."""
    if any(G.defaults.get(k) for k in ("FDC_FORCE_GENERIC", "FDC_NO_POLY", "FDC_NO_BLOCK")):
        pytest.skip("suite run under a forced path")
    N, R, nb = 65536, 4, 7
    x = noise(nb * (N - N // R), 77 + wt)
    for slots in (list(range(256)), [255, 0, 3, 128, 64, 200, 17]):
        chans = [(256 * c, 256, 0.88, 1.0) for c in slots]
        p = G.Pipeline(N, R, chans, windowtype=wt, max_blocks=nb)
        assert p.path() == 3
        outs = p.work(x)
        check = range(len(chans)) if len(chans) < 16 else (0, 1, 100, 127, 128, 254, 255)
        ref, _ = oracle.channelizer(N, R, wt, [chans[c] for c in check], x, nthreads=8)
        for i, c in enumerate(check):
            assert outs[c].size == nb * 192
            assert_close(outs[c], ref[i], "slot %d" % slots[c])
        q = G.Pipeline(N, R, chans, windowtype=wt, max_blocks=nb, flags=G.FDC_PIPE_NO_BLOCK)
        assert q.path() == 2
        for a, b in zip(outs, q.work(x)):
            assert_close(a, b, "one kernel vs two launches")
        p.reset()
        parts = [p.work(x[a * p.H:b * p.H]) for a, b in [(0, 3), (3, 4), (4, 7)]]
        for c in range(len(chans)):
            assert np.array_equal(np.concatenate([pp[c] for pp in parts]), outs[c])
    # an offset tiling at R = 4 takes the one-kernel form too (the window phase is a constant j^p per block: test_offset_tilings_at_relinvovl_4)
    off = G.Pipeline(N, R, [(256 * c + 37, 256, 0.88, 1.0) for c in range(8)], windowtype=1, max_blocks=2)
    assert off.path() == 3


@pytest.mark.parametrize("N,R,nb", [(16384, 2, 7), (16384, 2, 600), (32768, 2, 5), (32768, 2, 530), (16384, 4, 9), (32768, 4, 300)])
def test_one_kernel_path_other_block_lengths(oracle, N, R, nb):
    """The one-kernel form is a template on the number of 32-column passes: N = 16384 (64 slots, 2 passes) and N = 32768 (128
    slots, 4 passes) beside 65536 (l is data, python/FrequencyDomainChannelizer.py:323-327, and so is the
    block length).  All slots and a scattered subset; block counts below and above one round of workgroups; against the oracle
    (head and tail) and against the two-launch form on every sample; ragged calls bit for bit."""
    if any(G.defaults.get(k) for k in ("FDC_FORCE_GENERIC", "FDC_NO_POLY", "FDC_NO_BLOCK")):
        pytest.skip("suite run under a forced path")
    H, n1, lout = N - N // R, N // 256, 256 - 256 // R
    x = noise(nb * H, N // 256 + R + nb)
    rng = np.random.default_rng(N + nb)
    for slots in (list(range(n1)), [int(v) for v in rng.permutation(n1)[:9]]):
        chans = [(256 * c, 256, 0.88, 1.0) for c in slots]
        G.defaults["FDC_HOST_SUB"] = str(nb)
        try:
            p = G.Pipeline(N, R, chans, windowtype=1, max_blocks=nb)
            assert p.path() == 3
            outs = p.work(x)
        finally:
            G.defaults.pop("FDC_HOST_SUB", None)
        k = min(nb, 3)
        ref, _ = oracle.channelizer(N, R, 1, chans, x[:k * H], nthreads=8)
        for c in range(len(chans)):
            assert outs[c].size == nb * lout
            assert_close(outs[c][:k * lout], ref[c], "N %d slot %d head" % (N, slots[c]))
        if nb > k:
            t0 = nb - k
            ref2, _ = oracle.channelizer(N, R, 1, chans, x[t0 * H:], prefix=x[t0 * H - N // R:t0 * H], first_block=t0, nthreads=8)
            for c in range(len(chans)):
                assert_close(outs[c][t0 * lout:], ref2[c], "N %d slot %d tail" % (N, slots[c]))
        q = G.Pipeline(N, R, chans, windowtype=1, max_blocks=nb, flags=G.FDC_PIPE_NO_BLOCK)
        assert q.path() == 2
        for c, (a, b_) in enumerate(zip(outs, q.work(x))):
            assert_close(a, b_, "N %d slot %d vs two launches" % (N, slots[c]))
        p.reset()
        cuts = [(0, 1), (1, nb // 2), (nb // 2, nb)]
        parts = [p.work(x[a * H:b_ * H]) for a, b_ in cuts if b_ > a]
        for c in range(len(chans)):
            assert np.array_equal(np.concatenate([pp[c] for pp in parts]), outs[c])


@pytest.mark.parametrize("N", [16384, 32768])
def test_one_kernel_path_other_block_lengths_offsets_and_classes(oracle, N):
    """Offset tilings (f = 256 slot + r, odd r: the window phase follows the global block index) and plan classes at the other two
    block lengths of the one-kernel form."""
    if any(G.defaults.get(k) for k in ("FDC_FORCE_GENERIC", "FDC_NO_POLY", "FDC_NO_BLOCK")):
        pytest.skip("suite run under a forced path")
    R, nb = 2, 6
    H, n1 = N - N // R, N // 256
    x = noise(nb * H, 5 + n1)
    rng = np.random.default_rng(n1)
    for r in (1, 16, 37, 128, 255):
        slots = [int(v) for v in rng.permutation(n1 - 1)[:min(n1 - 1, 11)]]
        chans = [(256 * c + r, 256, 0.88, 1.0) for c in slots]
        p = G.Pipeline(N, R, chans, windowtype=1, max_blocks=nb)
        assert p.path() == 3
        outs = p.work(x)
        ref, _ = oracle.channelizer(N, R, 1, chans, x, nthreads=8)
        for c in range(len(chans)):
            assert_close(outs[c], ref[c], "N %d offset %d slot %d" % (N, r, slots[c]))
        p.reset()
        parts = [p.work(x[a * H:b_ * H]) for a, b_ in [(0, 1), (1, 4), (4, 6)]]
        for c in range(len(chans)):
            assert np.array_equal(np.concatenate([pp[c] for pp in parts]), outs[c])
    # a 2x oversampled bank (two tilings 128 bins apart) and the same slots under two windows: one launch per class
    s0 = [int(v) for v in rng.permutation(n1 - 1)[:n1 // 2]]
    plan = [(256 * c, 256, 0.88, 1.0) for c in s0] + [(256 * c + 128, 256, 0.88, 1.0) for c in s0] + [(256 * c, 256, 0.7, 0.9) for c in s0[:5]]
    # (round 5: the cost rule applies at these block lengths too, and prices three launches for 1.04 of the band above the spectrum path:
    # FDC_PIPE_WIDE_UNIFORM keeps the banks)
    assert G.Pipeline(N, R, plan, windowtype=2, max_blocks=nb).path() in (1, 3)
    p = G.Pipeline(N, R, plan, windowtype=2, max_blocks=nb, flags=G.FDC_PIPE_WIDE_UNIFORM)
    assert p.path() == 3
    outs = p.work(x)
    ref, _ = oracle.channelizer(N, R, 2, plan, x, nthreads=8)
    for c in range(len(plan)):
        assert_close(outs[c], ref[c], "N %d classes channel %d" % (N, c))


@pytest.mark.parametrize("N,R", [(32768, 2), (16384, 2), (32768, 4), (16384, 4)])
def test_split_plans_at_shorter_blocks(oracle, N, R):
    """Round 5: with the forward variant of the block kernel at N = 32768 / 16384 (k_blk256<P, ..., FWD>) a plan that is almost a bank is split there
    too (fdc_pipeline_path() = 4): banks of two widths on their block kernels, the rest on a partial spectrum.  Every channel against the oracle and
    against the spectrum path; ragged calls bit for bit; launch groups below the block-kernel threshold (the remainder's two-pass transform)."""
    if any(G.defaults.get(k) for k in ("FDC_FORCE_GENERIC", "FDC_NO_POLY", "FDC_NO_BLOCK")):
        pytest.skip("suite run under a forced path")
    nb = 9
    H, n1 = N - N // R, N // 256
    x = noise(nb * H, N // 7 + R)
    rng = np.random.default_rng(N + R)
    odd = [(N // 3 | 1, 512, 0.7, 0.9), (N // 2 + 33, 128, 0.88, 1.0), (2049, 1024, 0.6, 0.85), (N - 5000, 64, 0.5, 0.8), (31, 256, 0.88, 1.0)]
    plans = {"bank + five others": [(256 * int(c), 256, 0.88, 1.0) for c in rng.permutation(n1)[:n1 - 6]] + odd,
             "two widths + others": [(256 * c, 256, 0.88, 1.0) for c in range(n1 // 2)] + [(512 * c, 512, 0.88, 1.0) for c in range(n1 // 4, n1 // 2)] + odd[1:4]}
    for name, plan in plans.items():
        # (two banks + a remainder cost more than the spectrum path by the cost rule: FDC_PIPE_WIDE_UNIFORM keeps the banks, for the sake of the test)
        fl = G.FDC_PIPE_WIDE_UNIFORM if name.startswith("two widths") else 0
        p = G.Pipeline(N, R, plan, windowtype=1, max_blocks=nb, min_block_launch=1, flags=fl)
        assert p.path() == 4, (name, p.describe())
        outs = p.work(x)
        ref, _ = oracle.channelizer(N, R, 1, plan, x, nthreads=8)
        for c in range(len(plan)):
            if plan[c][1] != 256 or c % 7 == 0 or (plan[c][0] & 255):
                assert_close(outs[c], ref[c], "N %d %s: channel %d %s" % (N, name, c, plan[c]))
        q = G.Pipeline(N, R, plan, windowtype=1, max_blocks=nb, flags=G.FDC_PIPE_NO_POLY)
        assert q.path() == 1
        for c, (a, b_) in enumerate(zip(outs, q.work(x))):
            assert_close(a, b_, "N %d %s: channel %d vs the spectrum path" % (N, name, c))
        p.reset()
        parts = [p.work(x[a * H:b_ * H]) for a, b_ in [(0, 1), (1, 4), (4, 9)]]
        for c in range(len(plan)):
            assert np.array_equal(np.concatenate([pp[c] for pp in parts]), outs[c]), (name, c)
        t = G.Pipeline(N, R, plan, windowtype=1, max_blocks=nb, min_block_launch=96, flags=fl)
        assert t.path() == 4
        for c, (a, b_) in enumerate(zip(t.work(x), outs)):
            assert_close(a, b_, "N %d %s: channel %d with short launch groups" % (N, name, c))


def test_split_plans_classes_plus_remainder(oracle):
    """a plan that is ALMOST a bank — tilings of 256-bin channels plus a few channels of other widths, off every
    tiling, or a fourth tiling — is split: the tilings on the one-kernel form (one launch each), the remainder on the spectrum path
    over a partial spectrum that holds only what it reads (fdc_pipeline_path() = 4), where the cost rule says that beats the whole
    plan on the spectrum path.  Every channel against the oracle and against the spectrum path; ragged calls; short launch groups."""
    if any(G.defaults.get(k) for k in ("FDC_FORCE_GENERIC", "FDC_NO_POLY", "FDC_NO_BLOCK")):
        pytest.skip("suite run under a forced path")
    N, R, nb = 65536, 2, 6
    H = N - N // R
    x = noise(nb * H, 4711)
    rng = np.random.default_rng(9)
    bank = [(256 * int(c), 256, 0.88, 1.0) for c in rng.permutation(256)[:230]]
    odd = [(12345, 512, 0.7, 0.9), (40001, 128, 0.88, 1.0), (2049, 1024, 0.6, 0.85), (60000, 64, 0.5, 0.8), (31, 256, 0.88, 1.0), (50000, 2048, 0.9, 1.0)]
    plans = {"bank + six others": bank + odd,
             "interleaved": [ch for pair in zip(bank[:6], odd) for ch in pair] + bank[6:],
             "two tilings + others": bank + [(256 * int(c) + 128, 256, 0.88, 1.0) for c in rng.permutation(255)[:200]] + odd[:3]}
    for name, plan in plans.items():
        p = G.Pipeline(N, R, plan, windowtype=1, max_blocks=nb)
        assert p.path() == 4, name
        outs = p.work(x)
        ref, _ = oracle.channelizer(N, R, 1, plan, x, nthreads=8)
        for c in range(len(plan)):
            if plan[c][1] != 256 or c % 17 == 0 or (plan[c][0] & 255):
                assert_close(outs[c], ref[c], "%s: channel %d %s" % (name, c, plan[c]))
        q = G.Pipeline(N, R, plan, windowtype=1, max_blocks=nb, flags=G.FDC_PIPE_NO_POLY)
        assert q.path() == 1
        for c, (a, b_) in enumerate(zip(outs, q.work(x))):
            assert_close(a, b_, "%s: channel %d vs the spectrum path" % (name, c))
        p.reset()
        parts = [p.work(x[a * H:b_ * H]) for a, b_ in [(0, 1), (1, 4), (4, 6)]]
        for c in range(len(plan)):
            assert np.array_equal(np.concatenate([pp[c] for pp in parts]), outs[c]), (name, c)
        # launch groups below the block-kernel threshold: tiled kernels for the tiling (one on-grid class) and for the remainder's transform
        t = G.Pipeline(N, R, plan, windowtype=1, max_blocks=nb, min_block_launch=96)
        assert t.path() == 4
        for c, (a, b_) in enumerate(zip(t.work(x), outs)):
            assert_close(a, b_, "%s: channel %d on the tiled kernels" % (name, c))
    # the spectrum port (debug) of a split plan is the whole spectrum, and its channels are unchanged
    p = G.Pipeline(N, R, plans["bank + six others"], windowtype=1, max_blocks=nb, keep_spectrum=True)
    outs, spec = p.work(x, want_spectrum=True)
    ref, rspec = oracle.channelizer(N, R, 1, plans["bank + six others"], x, nthreads=8, want_spectrum=True)
    assert_close(spec, np.asarray(rspec).reshape(-1), "spectrum port")
    for c in (0, 229, 230, 235):
        assert_close(outs[c], ref[c])
    # half the band in 128-bin channels half a channel off their grid: since round 5 a second BANK (k_blknar), two launches (tests/test_plan_choice_gpu.py)
    half = [(256 * c, 256, 0.88, 1.0) if c % 2 == 0 else (256 * c + 64, 128, 0.88, 1.0) for c in range(256)]
    ph = G.Pipeline(N, R, half, windowtype=1, max_blocks=nb)
    assert ph.path() == 3 and "two launches" in ph.describe(), ph.describe()
    for c, (a, b_) in enumerate(zip(ph.work(x), G.Pipeline(N, R, half, windowtype=1, max_blocks=nb, flags=G.FDC_PIPE_NO_POLY).work(x))):
        assert_close(a, b_, "two banks of two widths: channel %d vs the spectrum path" % c)
    # where neither banks nor a split pay the plan stays on the spectrum path: bench.py --mixed (the 512-bin channels sit a quarter off their grid) ...
    mixed = [(256 * c, 256, 0.88, 1.0) for c in range(0, 256, 2)] + [(256 * c + 64, 128, 0.88, 1.0) for c in range(1, 256, 4)] + \
            [(256 * c - 128, 512, 0.88, 1.0) for c in range(3, 252, 4)]
    assert G.Pipeline(N, R, mixed, windowtype=1, max_blocks=2).path() == 1
    # ... and a handful of channels
    assert G.Pipeline(N, R, bank[:20] + odd[:1], windowtype=1, max_blocks=2).path() == 1


@pytest.mark.parametrize("N,L,R,wt", [(65536, 512, 2, 1), (65536, 128, 2, 1), (65536, 1024, 4, 2), (65536, 64, 2, 0), (65536, 2048, 2, 1),
                                      (16384, 512, 2, 1), (4096, 128, 4, 1), (262144, 1024, 2, 1), (32768, 2048, 8, 2)])
def test_uniform_banks_of_other_widths(oracle, N, L, R, wt):
    """Uniform banks whose channels are not 256 bins wide (every channel l = L on the L-bin grid, one window): the commutation of the
    l = 256 path does not depend on the width — stage 1 on the generic LDS core (k_p1g), stage 2 on k_p2g, G through memory, no spectrum
    (fdc_pipeline_path() = 2).  All slots and a scattered subset against the oracle and against the spectrum path; ragged calls."""
    if any(G.defaults.get(k) for k in ("FDC_FORCE_GENERIC", "FDC_NO_POLY")):
        pytest.skip("suite run under a forced path")
    nb = 5
    H, n1 = N - N // R, N // L
    x = noise(nb * H, L + R)
    rng = np.random.default_rng(N // L + R)
    for slots in ([int(v) for v in rng.permutation(n1)[:7]], list(range(n1)) if n1 <= 128 else [int(v) for v in rng.permutation(n1)[:100]]):
        chans = [(L * c, L, 0.88, 1.0) for c in slots]
        # every width on request; by default only where it measured faster than the spectrum path (l = 128)
        # (FDC_PIPE_NO_BLOCK: l = 512 at N = 65536, R = 2 has a block kernel of its own, tested below)
        # (round 6: N = 4096 has the one-launch form for this width, tests/test_fused4096_gpu.py; this test is about what FDC_PIPE_NO_FUSED leaves)
        nf = G.FDC_PIPE_NO_FUSED if N == 4096 else 0
        assert N != 4096 or G.defaults.get("FDC_NO_FUSED") or G.Pipeline(N, R, chans, windowtype=wt, max_blocks=nb).path() == 5
        p = G.Pipeline(N, R, chans, windowtype=wt, max_blocks=nb, chunk_blocks=3, flags=G.FDC_PIPE_WIDE_UNIFORM | G.FDC_PIPE_NO_BLOCK | nf)
        assert p.path() == 2
        assert (G.Pipeline(N, R, chans, windowtype=wt, max_blocks=nb, flags=G.FDC_PIPE_NO_BLOCK | nf).path() == 2) == (L == 128)
        outs = p.work(x)
        check = range(len(chans)) if len(chans) <= 16 else range(0, len(chans), max(1, len(chans) // 12))
        ref, _ = oracle.channelizer(N, R, wt, [chans[c] for c in check], x, nthreads=8)
        for i, c in enumerate(check):
            assert outs[c].size == nb * (L - L // R)
            assert_close(outs[c], ref[i], "N %d L %d slot %d" % (N, L, slots[c]))
        q = G.Pipeline(N, R, chans, windowtype=wt, max_blocks=nb, flags=G.FDC_PIPE_NO_POLY)
        assert q.path() in (0, 1)
        for c, (a, b_) in enumerate(zip(outs, q.work(x))):
            assert_close(a, b_, "N %d L %d slot %d vs the spectrum path" % (N, L, slots[c]))
        p.reset()
        parts = [p.work(x[a * H:b_ * H]) for a, b_ in [(0, 2), (2, 3), (3, 5)]]
        for c in range(len(chans)):
            assert np.array_equal(np.concatenate([pp[c] for pp in parts]), outs[c])
    # off the L-grid, two windows, a slot twice: not this path
    assert G.Pipeline(N, R, [(L * 1 + 1, L, 0.88, 1.0), (L * 3, L, 0.88, 1.0)], windowtype=wt, max_blocks=2, flags=G.FDC_PIPE_WIDE_UNIFORM).path() != 2
    assert G.Pipeline(N, R, [(L * 1, L, 0.7, 0.9), (L * 3, L, 0.88, 1.0)], windowtype=wt, max_blocks=2, flags=G.FDC_PIPE_WIDE_UNIFORM).path() != 2


@pytest.mark.parametrize("nslots,nb", [(128, 7), (128, 300), (9, 261), (1, 3), (128, 530)])
def test_one_kernel_path_for_512_bin_channels(oracle, nslots, nb):
    """l = 512 at N = 65536, R = 2 (fdc_block512.hip): the parities of a column's 512 rows as two 256-point columns in the lanes of
    a quad, joined by radix-2 layers through DPP.  Against the oracle (head and tail), against the generic two-launch form on every
    sample, block counts below and above one round of workgroups, ragged calls bit for bit, all three window shapes."""
    if any(G.defaults.get(k) for k in ("FDC_FORCE_GENERIC", "FDC_NO_POLY", "FDC_NO_BLOCK")):
        pytest.skip("suite run under a forced path")
    N, R, L = 65536, 2, 512
    H = N - N // R
    rng = np.random.default_rng(nslots * 7 + nb)
    slots = [int(v) for v in rng.permutation(128)[:nslots]]
    wt = nb % 3
    chans = [(L * c, L, 0.88, 1.0) for c in slots]
    x = noise(nb * H, 512 + nb)
    G.defaults["FDC_HOST_SUB"] = str(nb)
    try:
        p = G.Pipeline(N, R, chans, windowtype=wt, max_blocks=nb)
        assert p.path() == 3
        outs = p.work(x)
    finally:
        G.defaults.pop("FDC_HOST_SUB", None)
    k = min(nb, 3)
    ref, _ = oracle.channelizer(N, R, wt, chans, x[:k * H], nthreads=8)
    for c in range(len(chans)):
        assert outs[c].size == nb * 256
        assert_close(outs[c][:k * 256], ref[c], "slot %d head" % slots[c])
    if nb > k:
        t0 = nb - k
        ref2, _ = oracle.channelizer(N, R, wt, chans, x[t0 * H:], prefix=x[t0 * H - N // R:t0 * H], first_block=t0, nthreads=8)
        for c in range(len(chans)):
            assert_close(outs[c][t0 * 256:], ref2[c], "slot %d tail" % slots[c])
    q = G.Pipeline(N, R, chans, windowtype=wt, max_blocks=nb, flags=G.FDC_PIPE_NO_BLOCK | G.FDC_PIPE_WIDE_UNIFORM)
    assert q.path() == 2
    for c, (a, b_) in enumerate(zip(outs, q.work(x))):
        assert_close(a, b_, "slot %d vs the two-launch form" % slots[c])
    p.reset()
    cuts = [(0, 1), (1, nb // 2), (nb // 2, nb)]
    parts = [p.work(x[a * H:b_ * H]) for a, b_ in cuts if b_ > a]
    for c in range(len(chans)):
        assert np.array_equal(np.concatenate([pp[c] for pp in parts]), outs[c])


@pytest.mark.parametrize("L,nslots,nb", [(128, 512, 5), (128, 512, 300), (128, 37, 261), (128, 1, 3), (128, 511, 530),
                                         (64, 1024, 5), (64, 1024, 290), (64, 61, 261), (64, 1, 3), (64, 1023, 521)])
def test_one_kernel_path_for_narrow_channels(oracle, L, nslots, nb):
    """l = 128 and 64 at N = 65536, R = 2 (fdc_blocknarrow.hip): 256/l adjacent columns interleaved into one 256-point virtual column,
    separated and re-joined in registers; the last layer of the FFT over the slots between the lanes of a quad.  Against the oracle (head and
    tail; all slots of small banks, a spread of a full one with the slots at the seams of the layout), against the generic two-launch form on
    every sample, block counts below and above one round of workgroups, ragged calls bit for bit, all three window shapes."""
    if any(G.defaults.get(k) for k in ("FDC_FORCE_GENERIC", "FDC_NO_POLY", "FDC_NO_BLOCK")):
        pytest.skip("suite run under a forced path")
    N, R = 65536, 2
    H = N - N // R
    N1, lout = N // L, L // 2
    rng = np.random.default_rng(nslots * 11 + nb)
    slots = [int(v) for v in rng.permutation(N1)[:nslots]]
    wt = nb % 3
    chans = [(L * c, L, 0.88, 1.0) for c in slots]
    x = noise(nb * H, L + nb)
    G.defaults["FDC_HOST_SUB"] = str(nb)
    try:
        p = G.Pipeline(N, R, chans, windowtype=wt, max_blocks=nb)
        assert p.path() == 3
        outs = p.work(x)
    finally:
        G.defaults.pop("FDC_HOST_SUB", None)
    check = list(range(len(chans))) if len(chans) < 48 else sorted(set([0, 1, 2, len(chans) - 1] + [int(v) for v in rng.integers(0, len(chans), 24)]))
    # the slots next to the seams of the layout (k and k + 256 i sit in the lanes of one quad), when the bank has them
    for sl in (0, 1, 255, 256, 257, 511, 512, 513, 767, 768, 1023):
        if sl in slots and slots.index(sl) not in check:
            check.append(slots.index(sl))
    sub = [chans[c] for c in check]
    k = min(nb, 3)
    ref, _ = oracle.channelizer(N, R, wt, sub, x[:k * H], nthreads=8)
    for i, c in enumerate(check):
        assert outs[c].size == nb * lout
        assert_close(outs[c][:k * lout], ref[i], "l %d slot %d head" % (L, slots[c]))
    if nb > k:
        t0 = nb - k
        ref2, _ = oracle.channelizer(N, R, wt, sub, x[t0 * H:], prefix=x[t0 * H - N // R:t0 * H], first_block=t0, nthreads=8)
        for i, c in enumerate(check):
            assert_close(outs[c][t0 * lout:], ref2[i], "l %d slot %d tail" % (L, slots[c]))
    q = G.Pipeline(N, R, chans, windowtype=wt, max_blocks=nb, flags=G.FDC_PIPE_NO_BLOCK | G.FDC_PIPE_WIDE_UNIFORM)
    assert q.path() == 2
    for c, (a, b_) in enumerate(zip(outs, q.work(x))):
        assert_close(a, b_, "l %d slot %d vs the two-launch form" % (L, slots[c]))
    p.reset()
    cuts = [(0, 1), (1, nb // 2), (nb // 2, nb)]
    parts = [p.work(x[a * H:b_ * H]) for a, b_ in cuts if b_ > a]
    for c in range(len(chans)):
        assert np.array_equal(np.concatenate([pp[c] for pp in parts]), outs[c])


@pytest.mark.parametrize("wt", [0, 1])
def test_512_bin_block_kernel_at_relinvovl_4(oracle, wt):
    """l = 512 at R = 4 (the reference's default overlap): 384 of the 512 samples of every inverse transform are kept — 256 in the G
    registers, 128 through the per-workgroup scratch and a third run of stage 2.  All 128 slots and a subset, against the oracle and against the
    generic two-launch form on every sample; several workgroup rounds; ragged calls."""
    if any(G.defaults.get(k) for k in ("FDC_FORCE_GENERIC", "FDC_NO_POLY", "FDC_NO_BLOCK")):
        pytest.skip("suite run under a forced path")
    N, R, L, nb = 65536, 4, 512, 7
    H = N - N // R
    x = noise(nb * H, 99 + wt)
    for slots in (list(range(128)), [127, 0, 3, 64, 100, 17]):
        chans = [(L * c, L, 0.88, 1.0) for c in slots]
        p = G.Pipeline(N, R, chans, windowtype=wt, max_blocks=nb)
        assert p.path() == 3
        outs = p.work(x)
        check = range(len(chans)) if len(chans) < 16 else (0, 1, 63, 64, 126, 127)
        ref, _ = oracle.channelizer(N, R, wt, [chans[c] for c in check], x, nthreads=8)
        for i, c in enumerate(check):
            assert outs[c].size == nb * 384
            assert_close(outs[c], ref[i], "slot %d" % slots[c])
        q = G.Pipeline(N, R, chans, windowtype=wt, max_blocks=nb, flags=G.FDC_PIPE_NO_BLOCK | G.FDC_PIPE_WIDE_UNIFORM)
        assert q.path() == 2
        for a, b_ in zip(outs, q.work(x)):
            assert_close(a, b_, "block kernel vs two launches")
        p.reset()
        parts = [p.work(x[a * H:b_ * H]) for a, b_ in [(0, 3), (3, 4), (4, 7)]]
        for c in range(len(chans)):
            assert np.array_equal(np.concatenate([pp[c] for pp in parts]), outs[c])


@pytest.mark.parametrize("L,N,R,off,nslots,nb", [(512, 32768, 2, 0, 64, 7), (512, 32768, 2, 0, 5, 261), (512, 32768, 2, 256, 64, 9), (512, 32768, 4, 0, 64, 7),
                                                 (512, 32768, 4, 256, 11, 263), (512, 16384, 2, 0, 32, 7), (512, 16384, 2, 0, 3, 530), (512, 16384, 2, 256, 32, 261), (512, 16384, 4, 0, 32, 9), (512, 16384, 4, 256, 7, 263),
                                                 (1024, 32768, 2, 0, 32, 7), (1024, 32768, 2, 0, 5, 261), (1024, 32768, 2, 512, 32, 9), (1024, 32768, 4, 0, 32, 7),
                                                 (1024, 32768, 4, 512, 11, 263), (1024, 16384, 2, 0, 16, 7), (1024, 16384, 2, 0, 3, 530), (1024, 16384, 2, 512, 16, 261), (1024, 16384, 4, 0, 16, 9), (1024, 16384, 4, 512, 5, 263)])
def test_wide_block_kernels_at_shorter_blocks(oracle, L, N, R, off, nslots, nb):
    """k_blk512<P> / k_blk1024<P> (round 5): the 512- and 1024-bin block kernels at N = 8192 P, P = 4 and 2 passes (N = 32768, R = 2 and 4;
    N = 16384, R = 2) — banks on the grid and half a channel off it.  Against the oracle (head and tail), against the spectrum path on every
    sample, block counts below and above one round of workgroups, ragged calls bit for bit, the three window shapes."""
    if any(G.defaults.get(k) for k in ("FDC_FORCE_GENERIC", "FDC_NO_POLY", "FDC_NO_BLOCK")):
        pytest.skip("suite run under a forced path")
    H, n1, lout = N - N // R, N // L, L - L // R
    rng = np.random.default_rng(N // L + 3 * R + nb + off)
    slots = [int(v) for v in rng.permutation(n1 - (1 if off else 0))[:nslots]]
    wt = nb % 3
    chans = [(L * c + off, L, 0.8, 0.95) for c in slots]
    x = noise(nb * H, N // 100 + nb)
    G.defaults["FDC_HOST_SUB"] = str(nb)
    try:
        p = G.Pipeline(N, R, chans, windowtype=wt, max_blocks=nb)
        assert p.path() == 3 and ("k_blk%d" % L) in p.describe()
        outs = p.work(x)
    finally:
        G.defaults.pop("FDC_HOST_SUB", None)
    check = list(range(len(chans))) if len(chans) <= 12 else sorted(set([0, 1, len(chans) - 1] + [int(v) for v in rng.integers(0, len(chans), 8)]))
    sub = [chans[c] for c in check]
    k = min(nb, 3)
    ref, _ = oracle.channelizer(N, R, wt, sub, x[:k * H], nthreads=8)
    for i, c in enumerate(check):
        assert outs[c].size == nb * lout
        assert_close(outs[c][:k * lout], ref[i], "l %d N %d slot %d head" % (L, N, slots[c]))
    if nb > k:
        t0 = nb - k
        ref2, _ = oracle.channelizer(N, R, wt, sub, x[t0 * H:], prefix=x[t0 * H - N // R:t0 * H], first_block=t0, nthreads=8)
        for i, c in enumerate(check):
            assert_close(outs[c][t0 * lout:], ref2[i], "l %d N %d slot %d tail" % (L, N, slots[c]))
    q = G.Pipeline(N, R, chans, windowtype=wt, max_blocks=nb, flags=G.FDC_PIPE_NO_POLY)
    assert q.path() in (0, 1)
    for c, (a, b_) in enumerate(zip(outs, q.work(x))):
        assert_close(a, b_, "l %d N %d slot %d vs the spectrum path" % (L, N, slots[c]))
    p.reset()
    cuts = [(0, 1), (1, nb // 2), (nb // 2, nb)]
    parts = [p.work(x[a * H:b_ * H]) for a, b_ in cuts if b_ > a]
    for c in range(len(chans)):
        assert np.array_equal(np.concatenate([pp[c] for pp in parts]), outs[c])


@pytest.mark.parametrize("L,N,R,off,nslots,nb", [(128, 32768, 2, 0, 256, 7), (128, 32768, 2, 64, 40, 261), (128, 32768, 2, 32, 255, 9), (128, 32768, 4, 0, 256, 7),
                                                 (128, 32768, 4, 96, 11, 263), (128, 16384, 2, 0, 128, 7), (128, 16384, 2, 64, 3, 530), (128, 16384, 4, 0, 128, 9),
                                                 (128, 16384, 4, 32, 17, 261),
                                                 (64, 32768, 2, 0, 512, 7), (64, 32768, 2, 32, 40, 261), (64, 32768, 2, 16, 511, 9), (64, 32768, 4, 0, 512, 7),
                                                 (64, 32768, 4, 48, 11, 263), (64, 16384, 2, 0, 256, 7), (64, 16384, 2, 32, 3, 530), (64, 16384, 4, 0, 256, 9),
                                                 (64, 16384, 4, 16, 17, 261)])
def test_narrow_block_kernel_at_shorter_blocks(oracle, L, N, R, off, nslots, nb):
    """k_blknar<S, ..., P> (round 5): the narrow-channel block kernel at N = 8192 P, P = 4 and 2 passes of 32 virtual columns (N = 32768 and
    16384, R = 2 and 4): banks on the grid, half and a quarter / three quarters of a channel off it.  A trip of stage 2 then holds more 64-row
    blocks than a run has (waves without a row).  Against the oracle (head and tail; the slots at the seams of the layout), against the
    spectrum path on every sample, several workgroup rounds, ragged calls bit for bit, the three window shapes."""
    if any(G.defaults.get(k) for k in ("FDC_FORCE_GENERIC", "FDC_NO_POLY", "FDC_NO_BLOCK")):
        pytest.skip("suite run under a forced path")
    H, n1, lout = N - N // R, N // L, L - L // R
    nv = N // 256
    rng = np.random.default_rng(N // L + 3 * R + nb + off)
    slots = [int(v) for v in rng.permutation(n1 - (1 if off else 0))[:nslots]]
    wt = nb % 3
    chans = [(L * c + off, L, 0.8, 0.95) for c in slots]
    x = noise(nb * H, N // 100 + nb + L)
    G.defaults["FDC_HOST_SUB"] = str(nb)
    try:
        p = G.Pipeline(N, R, chans, windowtype=wt, max_blocks=nb)
        assert p.path() == 3 and "k_blknar" in p.describe()
        outs = p.work(x)
    finally:
        G.defaults.pop("FDC_HOST_SUB", None)
    check = list(range(len(chans))) if len(chans) <= 12 else sorted(set([0, 1, len(chans) - 1] + [int(v) for v in rng.integers(0, len(chans), 8)]))
    for sl in (0, 1, nv - 1, nv, nv + 1, 2 * nv - 1, 2 * nv, 3 * nv, n1 - 1):      # k and k + 32 P i sit in the lanes of one quad
        if sl in slots and slots.index(sl) not in check:
            check.append(slots.index(sl))
    sub = [chans[c] for c in check]
    k = min(nb, 3)
    ref, _ = oracle.channelizer(N, R, wt, sub, x[:k * H], nthreads=8)
    for i, c in enumerate(check):
        assert outs[c].size == nb * lout
        assert_close(outs[c][:k * lout], ref[i], "l %d N %d slot %d head" % (L, N, slots[c]))
    if nb > k:
        t0 = nb - k
        ref2, _ = oracle.channelizer(N, R, wt, sub, x[t0 * H:], prefix=x[t0 * H - N // R:t0 * H], first_block=t0, nthreads=8)
        for i, c in enumerate(check):
            assert_close(outs[c][t0 * lout:], ref2[i], "l %d N %d slot %d tail" % (L, N, slots[c]))
    q = G.Pipeline(N, R, chans, windowtype=wt, max_blocks=nb, flags=G.FDC_PIPE_NO_POLY)
    assert q.path() in (0, 1)
    for c, (a, b_) in enumerate(zip(outs, q.work(x))):
        assert_close(a, b_, "l %d N %d slot %d vs the spectrum path" % (L, N, slots[c]))
    p.reset()
    cuts = [(0, 1), (1, nb // 2), (nb // 2, nb)]
    parts = [p.work(x[a * H:b_ * H]) for a, b_ in cuts if b_ > a]
    for c in range(len(chans)):
        assert np.array_equal(np.concatenate([pp[c] for pp in parts]), outs[c])


@pytest.mark.parametrize("L,wt", [(128, 0), (128, 1), (64, 0), (64, 2)])
def test_narrow_block_kernel_at_relinvovl_4(oracle, L, wt):
    """l = 128 / 64 at R = 4 (the reference's default overlap): three quarters of every inverse transform are kept — the rows t >= 128 of a
    virtual column in the G registers, the rows 64 .. 127 through the per-workgroup scratch and a second run of stage 2.  The full bank and a
    subset, against the oracle and against the generic two-launch form on every sample; several workgroup rounds; ragged calls."""
    if any(G.defaults.get(k) for k in ("FDC_FORCE_GENERIC", "FDC_NO_POLY", "FDC_NO_BLOCK")):
        pytest.skip("suite run under a forced path")
    N, R = 65536, 4
    H = N - N // R
    N1, lout = N // L, 3 * L // 4
    for nb, slots in ((7, list(range(N1))), (263, [N1 - 1, 0, 3, 256, 257, N1 // 2 + 5, 100, 17])):
        chans = [(L * c, L, 0.88, 1.0) for c in slots]
        x = noise(nb * H, 77 + wt + L)
        p = G.Pipeline(N, R, chans, windowtype=wt, max_blocks=nb)
        assert p.path() == 3
        outs = p.work(x)
        check = range(len(chans)) if len(chans) < 16 else (0, 1, 255, 256, 257, N1 // 2, N1 - 2, N1 - 1)
        k = min(nb, 4)
        ref, _ = oracle.channelizer(N, R, wt, [chans[c] for c in check], x[:k * H], nthreads=8)
        t0 = nb - k
        ref2, _ = oracle.channelizer(N, R, wt, [chans[c] for c in check], x[t0 * H:], prefix=x[t0 * H - N // R:t0 * H], first_block=t0, nthreads=8)
        for i, c in enumerate(check):
            assert outs[c].size == nb * lout
            assert_close(outs[c][:k * lout], ref[i], "l %d slot %d head" % (L, slots[c]))
            assert_close(outs[c][t0 * lout:], ref2[i], "l %d slot %d tail" % (L, slots[c]))
        q = G.Pipeline(N, R, chans, windowtype=wt, max_blocks=nb, flags=G.FDC_PIPE_NO_BLOCK | G.FDC_PIPE_WIDE_UNIFORM)
        assert q.path() == 2
        for a, b_ in zip(outs, q.work(x)):
            assert_close(a, b_, "block kernel vs two launches")
        p.reset()
        parts = [p.work(x[a * H:b_ * H]) for a, b_ in [(0, 3), (3, 4), (4, nb)]]
        for c in range(len(chans)):
            assert np.array_equal(np.concatenate([pp[c] for pp in parts]), outs[c])


@pytest.mark.parametrize("nslots,nb", [(64, 7), (64, 300), (9, 261), (1, 3), (63, 530)])
def test_one_kernel_path_for_1024_bin_channels(oracle, nslots, nb):
    """l = 1024 at N = 65536, R = 2 (fdc_block1024.hip): the four phases of a column's 1024 rows as four 256-point columns in the lanes of a
    quad, joined by radix-4 layers through DPP.  Against the oracle (head and tail), against the generic two-launch form on every sample,
    block counts below and above one round of workgroups, ragged calls bit for bit, all three window shapes."""
    if any(G.defaults.get(k) for k in ("FDC_FORCE_GENERIC", "FDC_NO_POLY", "FDC_NO_BLOCK")):
        pytest.skip("suite run under a forced path")
    N, R, L = 65536, 2, 1024
    H = N - N // R
    rng = np.random.default_rng(nslots * 13 + nb)
    slots = [int(v) for v in rng.permutation(64)[:nslots]]
    wt = nb % 3
    chans = [(L * c, L, 0.88, 1.0) for c in slots]
    x = noise(nb * H, 1024 + nb)
    G.defaults["FDC_HOST_SUB"] = str(nb)
    try:
        # banks of fewer than 3 channels take the spectrum path unless asked (the kernel's cost does not depend on the number of channels;
        # csrc/fdc_plan_cost.hpp: 0.209 ms per 1024 blocks against 0.20 + 0.24 x the band read)
        assert G.Pipeline(N, R, chans, windowtype=wt, max_blocks=nb).path() == (3 if nslots >= 3 else 1)
        p = G.Pipeline(N, R, chans, windowtype=wt, max_blocks=nb, flags=G.FDC_PIPE_WIDE_UNIFORM)
        assert p.path() == 3
        outs = p.work(x)
    finally:
        G.defaults.pop("FDC_HOST_SUB", None)
    k = min(nb, 3)
    check = range(len(chans)) if len(chans) < 12 else (0, 1, 2, 31, 32, len(chans) - 2, len(chans) - 1)
    sub = [chans[c] for c in check]
    ref, _ = oracle.channelizer(N, R, wt, sub, x[:k * H], nthreads=8)
    for i, c in enumerate(check):
        assert outs[c].size == nb * 512
        assert_close(outs[c][:k * 512], ref[i], "slot %d head" % slots[c])
    if nb > k:
        t0 = nb - k
        ref2, _ = oracle.channelizer(N, R, wt, sub, x[t0 * H:], prefix=x[t0 * H - N // R:t0 * H], first_block=t0, nthreads=8)
        for i, c in enumerate(check):
            assert_close(outs[c][t0 * 512:], ref2[i], "slot %d tail" % slots[c])
    q = G.Pipeline(N, R, chans, windowtype=wt, max_blocks=nb, flags=G.FDC_PIPE_NO_BLOCK | G.FDC_PIPE_WIDE_UNIFORM)
    assert q.path() == 2
    for c, (a, b_) in enumerate(zip(outs, q.work(x))):
        assert_close(a, b_, "slot %d vs the two-launch form" % slots[c])
    p.reset()
    cuts = [(0, 1), (1, nb // 2), (nb // 2, nb)]
    parts = [p.work(x[a * H:b_ * H]) for a, b_ in cuts if b_ > a]
    for c in range(len(chans)):
        assert np.array_equal(np.concatenate([pp[c] for pp in parts]), outs[c])


@pytest.mark.parametrize("L,R,nb", [(128, 2, 261), (64, 2, 37), (512, 2, 270), (1024, 2, 261), (128, 4, 9), (64, 4, 261), (512, 4, 7), (1024, 4, 9)])
def test_banks_half_a_channel_higher(oracle, L, R, nb):
    """Banks centred on multiples of l (f = l slot + l/2) on the block kernels of the widths other than 256: the block modulated by
    exp(-2 pi i (l/2) n / N) moves every column's spectrum by half its length, which the kernels absorb in their tables (and one sign): the full
    bank (all slots but the last, which would wrap) and a subset; against the oracle, against the spectrum path on every sample, ragged calls
    (short launch groups included: this form has no two-launch fallback) bit for bit."""
    if any(G.defaults.get(k) for k in ("FDC_FORCE_GENERIC", "FDC_NO_POLY", "FDC_NO_BLOCK")):
        pytest.skip("suite run under a forced path")
    N = 65536
    H = N - N // R
    N1, lout = N // L, L - L // R
    rng = np.random.default_rng(L + R)
    wt = (L // 64 + R) % 3
    x = noise(nb * H, 5 * L + R)
    for slots in (list(range(N1 - 1)), sorted(int(v) for v in rng.permutation(N1 - 1)[:max(30, N1 // 9)])):
        chans = [(L * c + L // 2, L, 0.88, 1.0) for c in slots]
        p = G.Pipeline(N, R, chans, windowtype=wt, max_blocks=nb)
        assert p.path() == 3
        outs = p.work(x)
        check = sorted(set([0, 1, len(chans) // 2, len(chans) - 1] + [int(v) for v in rng.integers(0, len(chans), 6)]))
        for sl in (0, 255, 256, 511, 767):
            if sl in slots and slots.index(sl) not in check:
                check.append(slots.index(sl))
        sub = [chans[c] for c in check]
        k = min(nb, 3)
        ref, _ = oracle.channelizer(N, R, wt, sub, x[:k * H], nthreads=8)
        t0 = nb - k
        ref2, _ = oracle.channelizer(N, R, wt, sub, x[t0 * H:], prefix=x[t0 * H - N // R:t0 * H], first_block=t0, nthreads=8)
        for i, c in enumerate(check):
            assert outs[c].size == nb * lout
            assert_close(outs[c][:k * lout], ref[i], "l %d slot %d head" % (L, slots[c]))
            assert_close(outs[c][t0 * lout:], ref2[i], "l %d slot %d tail" % (L, slots[c]))
        q = G.Pipeline(N, R, chans, windowtype=wt, max_blocks=nb, flags=G.FDC_PIPE_NO_BLOCK)
        assert q.path() == 1
        for c, (a, b_) in enumerate(zip(outs, q.work(x))):
            assert_close(a, b_, "l %d slot %d vs the spectrum path" % (L, slots[c]))
        p.reset()
        cuts = [(0, 1), (1, 3), (3, max(3, nb // 2)), (max(3, nb // 2), nb)]
        parts = [p.work(x[a * H:b_ * H]) for a, b_ in cuts if b_ > a]
        for c in range(len(chans)):
            assert np.array_equal(np.concatenate([pp[c] for pp in parts]), outs[c])


def test_hier_block_with_a_bank_centred_on_multiples_of_the_channel_width(oracle):
    """The hier block face with 511 channels of 0.8/512 of the band centred on k/512 (user frequencies k/512 - 0.5): the reference's parameter
    derivation gives l = 128 at f = 128 k - 64, a bank half a channel off the 128-bin grid, which takes the narrow-channel block kernel."""
    if any(G.defaults.get(k) for k in ("FDC_FORCE_GENERIC", "FDC_NO_POLY", "FDC_NO_BLOCK")):
        pytest.skip("suite run under a forced path")
    N, R, C = 65536, 2, 512
    user = [[k / C - 0.5, 0.8 / C] for k in range(1, C)]
    fdc = G.FrequencyDomainChannelizer(8, 1, N, R, user, None, 6.0, 1.0, 0.0, 'normalized', 1,
                                       False, False, "", False, None, 10.0, 0.005, 1, 0.2, 0, 0, 128, 128, False, max_blocks=8)
    assert [cp[:3] for cp in fdc.channel_params[:3]] == [(64, 128, 64), (192, 128, 64), (320, 128, 64)]
    assert fdc.pipeline.path() == 3
    assert "k_blknar, l = 128, bank of 511 half a channel off the grid" in fdc.pipeline.describe()
    x = noise(8 * fdc.inpblocklen, 4242)
    ports = fdc.work(x)
    check = [0, 1, 255, 256, 509, 510]
    plan = [tuple(fdc.channel_params[c][i] for i in (0, 1, 3, 4)) for c in check]
    ref, _ = oracle.channelizer(N, R, 1, plan, x, nthreads=8)
    for i, c in enumerate(check):
        assert_close(ports[c], ref[i], "port %d" % c)


@pytest.mark.parametrize("wt", [0, 1])
def test_1024_bin_block_kernel_at_relinvovl_4(oracle, wt):
    """l = 1024 at R = 4 (the reference's default overlap): 768 of the 1024 samples of every inverse transform are kept — 512 in the G registers, 256
    through the per-workgroup scratch and a second run of stage 2.  The full bank and a subset, against the oracle and against the generic two-launch
    form on every sample; several workgroup rounds; ragged calls."""
    if any(G.defaults.get(k) for k in ("FDC_FORCE_GENERIC", "FDC_NO_POLY", "FDC_NO_BLOCK")):
        pytest.skip("suite run under a forced path")
    N, R, L = 65536, 4, 1024
    H = N - N // R
    for nb, slots in ((7, list(range(64))), (263, [63, 0, 3, 32, 33, 50, 17])):
        x = noise(nb * H, 199 + wt + nb)
        chans = [(L * c, L, 0.88, 1.0) for c in slots]
        p = G.Pipeline(N, R, chans, windowtype=wt, max_blocks=nb)
        assert p.path() == 3
        outs = p.work(x)
        check = range(len(chans)) if len(chans) < 16 else (0, 1, 31, 32, 62, 63)
        k = min(nb, 4)
        sub = [chans[c] for c in check]
        ref, _ = oracle.channelizer(N, R, wt, sub, x[:k * H], nthreads=8)
        t0 = nb - k
        ref2, _ = oracle.channelizer(N, R, wt, sub, x[t0 * H:], prefix=x[t0 * H - N // R:t0 * H], first_block=t0, nthreads=8)
        for i, c in enumerate(check):
            assert outs[c].size == nb * 768
            assert_close(outs[c][:k * 768], ref[i], "slot %d head" % slots[c])
            assert_close(outs[c][t0 * 768:], ref2[i], "slot %d tail" % slots[c])
        q = G.Pipeline(N, R, chans, windowtype=wt, max_blocks=nb, flags=G.FDC_PIPE_NO_BLOCK | G.FDC_PIPE_WIDE_UNIFORM)
        assert q.path() == 2
        for a, b_ in zip(outs, q.work(x)):
            assert_close(a, b_, "block kernel vs two launches")
        p.reset()
        parts = [p.work(x[a * H:b_ * H]) for a, b_ in [(0, 3), (3, 4), (4, nb)]]
        for c in range(len(chans)):
            assert np.array_equal(np.concatenate([pp[c] for pp in parts]), outs[c])


@pytest.mark.parametrize("N,R,nb", [(65536, 2, 261), (65536, 4, 261), (32768, 2, 9), (32768, 4, 300), (16384, 2, 530), (16384, 4, 7)])
def test_256_bin_bank_half_a_slot_higher(oracle, N, R, nb):
    """Channels of 256 bins centred on multiples of 256 (f = 256 slot + 128) on the block kernel's HALF form: the on-grid kernel with its tables read
    at k2 ^ 128 and no ifftshift — at relinvovl 4 too (the general offset form is R = 2 only).  Against the oracle, against the spectrum path on every
    sample, ragged calls bit for bit."""
    if any(G.defaults.get(k) for k in ("FDC_FORCE_GENERIC", "FDC_NO_POLY", "FDC_NO_BLOCK")):
        pytest.skip("suite run under a forced path")
    H = N - N // R
    N1, lout = N // 256, 256 - 256 // R
    rng = np.random.default_rng(N // 256 + R)
    wt = (N // 16384 + R) % 3
    x = noise(nb * H, N // 64 + R)
    for slots in (list(range(N1 - 1)), sorted(int(v) for v in rng.permutation(N1 - 1)[:N1 // 3])):
        chans = [(256 * c + 128, 256, 0.88, 1.0) for c in slots]
        p = G.Pipeline(N, R, chans, windowtype=wt, max_blocks=nb)
        assert p.path() == 3
        outs = p.work(x)
        check = sorted(set([0, 1, len(chans) // 2, len(chans) - 1] + [int(v) for v in rng.integers(0, len(chans), 5)]))
        sub = [chans[c] for c in check]
        k = min(nb, 3)
        ref, _ = oracle.channelizer(N, R, wt, sub, x[:k * H], nthreads=8)
        t0 = nb - k
        ref2, _ = oracle.channelizer(N, R, wt, sub, x[t0 * H:], prefix=x[t0 * H - N // R:t0 * H], first_block=t0, nthreads=8)
        for i, c in enumerate(check):
            assert outs[c].size == nb * lout
            assert_close(outs[c][:k * lout], ref[i], "N %d R %d slot %d head" % (N, R, slots[c]))
            assert_close(outs[c][t0 * lout:], ref2[i], "N %d R %d slot %d tail" % (N, R, slots[c]))
        q = G.Pipeline(N, R, chans, windowtype=wt, max_blocks=nb, flags=G.FDC_PIPE_NO_POLY)
        assert q.path() in (0, 1)
        for c, (a, b_) in enumerate(zip(outs, q.work(x))):
            assert_close(a, b_, "slot %d vs the spectrum path" % slots[c])
        p.reset()
        cuts = [(0, 1), (1, 3), (3, max(3, nb // 2)), (max(3, nb // 2), nb)]
        parts = [p.work(x[a * H:b_ * H]) for a, b_ in cuts if b_ > a]
        for c in range(len(chans)):
            assert np.array_equal(np.concatenate([pp[c] for pp in parts]), outs[c])


@pytest.mark.parametrize("N,r,nslots", [(65536, 1, 255), (65536, 2, 40), (65536, 3, 255), (65536, 37, 100), (65536, 130, 255), (65536, 255, 9),
                                        (32768, 5, 127), (16384, 66, 63)])
def test_offset_tilings_at_relinvovl_4(oracle, N, r, nslots):
    """Tilings off the 256-bin grid at R = 4 (the reference's default overlap): the window phase counter (block * (f mod 4)) mod 4 runs
    (lib/phase_shifting_windowing_vcc_impl.cc:57-58,82) and phase p of the window is the window times j^p — a constant per block in the block
    kernel.  Every residue of r mod 4; against the oracle, against the spectrum path, ragged calls starting at every block index mod 4."""
    if any(G.defaults.get(k) for k in ("FDC_FORCE_GENERIC", "FDC_NO_POLY", "FDC_NO_BLOCK")):
        pytest.skip("suite run under a forced path")
    R, nb = 4, 11
    H = N - N // R
    rng = np.random.default_rng(r * 17 + nslots)
    slots = [int(v) for v in rng.permutation(N // 256 - 1)[:nslots]]
    chans = [(256 * c + r, 256, 0.88, 1.0) for c in slots]
    x = noise(nb * H, 400 + r)
    p = G.Pipeline(N, R, chans, windowtype=1, max_blocks=nb)
    assert p.path() == 3
    outs = p.work(x)
    ref, _ = oracle.channelizer(N, R, 1, chans, x, nthreads=8)
    for c in range(0, len(chans), max(1, len(chans) // 12)):
        assert outs[c].size == nb * 192
        assert_close(outs[c], ref[c], "slot %d offset %d" % (slots[c], r))
    q = G.Pipeline(N, R, chans, windowtype=1, max_blocks=nb, flags=G.FDC_PIPE_NO_POLY)
    assert q.path() in (0, 1)
    for a, b_ in zip(outs, q.work(x)):
        assert_close(a, b_)
    p.reset()
    parts = [p.work(x[a * H:b_ * H]) for a, b_ in [(0, 1), (1, 3), (3, 6), (6, 11)]]
    for c in range(len(chans)):
        assert np.array_equal(np.concatenate([q_[c] for q_ in parts]), outs[c])


@pytest.mark.parametrize("C,R,nb", [(512, 2, 261), (1024, 2, 9), (128, 2, 261), (512, 4, 7), (128, 4, 263)])
def test_centred_bank_through_the_parameter_derivation_is_two_banks(oracle, C, R, nb):
    """C channels of 0.8/C of the band centred on k/C, k = 0 .. C-1, through get_opt_channelparams (python/FrequencyDomainChannelizer.py:322-345):
    l = 65536/C at f = l k - l/2, except channel 0, whose slice wraps below zero and is clamped to the top of the band — on the l-bin grid.  A bank half a
    channel off the grid plus one channel on it: two launches of the width's block kernel instead of the spectrum path.  Channel 0 and its neighbours
    against the oracle, every sample against the spectrum path, ragged calls bit for bit."""
    if any(G.defaults.get(k) for k in ("FDC_FORCE_GENERIC", "FDC_NO_POLY", "FDC_NO_BLOCK")):
        pytest.skip("suite run under a forced path")
    N = 65536
    H = N - N // R
    prm = [G.get_opt_channelparams(N, R, (k / C) % 1.0, 0.8 / C) for k in range(C)]
    L = prm[0][1]
    assert L == N // C and prm[0][0] == N - L and all(p_[0] == L * k - L // 2 and p_[1] == L for k, p_ in enumerate(prm) if k)
    chans = [(f, l, pb, sb) for (f, l, _lo, pb, sb) in prm]
    lout = L - L // R
    x = noise(nb * H, 9000 + C + R)
    p = G.Pipeline(N, R, chans, windowtype=1, max_blocks=nb)
    assert p.path() == 3
    assert "bank of %d half a channel off the grid + bank of 1 on the grid (two launches)" % (C - 1) in p.describe()
    outs = p.work(x)
    check = [0, 1, 2, C // 2, C - 2, C - 1]
    sub = [chans[c] for c in check]
    k = min(nb, 3)
    ref, _ = oracle.channelizer(N, R, 1, sub, x[:k * H], nthreads=8)
    t0 = nb - k
    ref2, _ = oracle.channelizer(N, R, 1, sub, x[t0 * H:], prefix=x[t0 * H - N // R:t0 * H], first_block=t0, nthreads=8)
    for i, c in enumerate(check):
        assert outs[c].size == nb * lout
        assert_close(outs[c][:k * lout], ref[i], "channel %d head" % c)
        assert_close(outs[c][t0 * lout:], ref2[i], "channel %d tail" % c)
    q = G.Pipeline(N, R, chans, windowtype=1, max_blocks=nb, flags=G.FDC_PIPE_NO_BLOCK)
    assert q.path() == 1
    for c, (a, b_) in enumerate(zip(outs, q.work(x))):
        assert_close(a, b_, "channel %d vs the spectrum path" % c)
    p.reset()
    cuts = [(0, 1), (1, 3), (3, max(3, nb // 2)), (max(3, nb // 2), nb)]
    parts = [p.work(x[a * H:b_ * H]) for a, b_ in cuts if b_ > a]
    for c in range(len(chans)):
        assert np.array_equal(np.concatenate([pp[c] for pp in parts]), outs[c])


@pytest.mark.parametrize("C,R,nb", [(256, 2, 261), (1024, 2, 9), (256, 4, 7)])
def test_centred_bank_of_half_overlapping_channels(oracle, C, R, nb):
    """C channels of 1/C of the band (no gaps) centred on k/C: the derivation doubles the slice (l = 2 N / C: neighbours overlap by half), the
    channels alternate between the l-bin grid and half a channel off it — two banks — and channel 0, wrapped and clamped, lands on channel C - 1's slice:
    computed once, copied.  Against the oracle, against the spectrum path, ragged calls."""
    if any(G.defaults.get(k) for k in ("FDC_FORCE_GENERIC", "FDC_NO_POLY", "FDC_NO_BLOCK")):
        pytest.skip("suite run under a forced path")
    N = 65536
    H = N - N // R
    prm = [G.get_opt_channelparams(N, R, (k / C) % 1.0, 1.0 / C) for k in range(C)]
    L = prm[0][1]
    assert L == 2 * N // C and prm[0][0] == prm[C - 1][0] == N - L
    chans = [(f, l, pb, sb) for (f, l, _lo, pb, sb) in prm]
    lout = L - L // R
    x = noise(nb * H, 9100 + C + R)
    p = G.Pipeline(N, R, chans, windowtype=1, max_blocks=nb)
    assert p.path() == 3 and "two launches" in p.describe() and "1 copies" in p.describe()
    outs = p.work(x)
    assert np.array_equal(outs[0], outs[C - 1])
    check = [0, 1, 2, 3, C // 2, C - 2, C - 1]
    sub = [chans[c] for c in check]
    k = min(nb, 3)
    ref, _ = oracle.channelizer(N, R, 1, sub, x[:k * H], nthreads=8)
    t0 = nb - k
    ref2, _ = oracle.channelizer(N, R, 1, sub, x[t0 * H:], prefix=x[t0 * H - N // R:t0 * H], first_block=t0, nthreads=8)
    for i, c in enumerate(check):
        assert outs[c].size == nb * lout
        assert_close(outs[c][:k * lout], ref[i], "channel %d head" % c)
        assert_close(outs[c][t0 * lout:], ref2[i], "channel %d tail" % c)
    q = G.Pipeline(N, R, chans, windowtype=1, max_blocks=nb, flags=G.FDC_PIPE_NO_BLOCK)
    assert q.path() == 1
    for c, (a, b_) in enumerate(zip(outs, q.work(x))):
        assert_close(a, b_, "channel %d vs the spectrum path" % c)
    p.reset()
    cuts = [(0, 1), (1, 3), (3, max(3, nb // 2)), (max(3, nb // 2), nb)]
    parts = [p.work(x[a * H:b_ * H]) for a, b_ in cuts if b_ > a]
    for c in range(len(chans)):
        assert np.array_equal(np.concatenate([pp[c] for pp in parts]), outs[c])


@pytest.mark.parametrize("C,R,nb", [(1024, 2, 261), (2048, 2, 9), (1024, 4, 7), (2048, 4, 261)])
def test_bank_of_half_overlapping_narrow_channels_between_the_raster_points(oracle, C, R, nb):
    """C channels of 1/C of the band centred on (k + 1/2)/C: doubled slices (l = 2 N / C = 128 or 64) at f = (l/2)(k - 1/2) — a quarter and three quarters of a
    channel off the l-bin grid, alternately: two banks of the narrow-channel kernel's ROT forms (the virtual column's spectrum moved by whole registers).
    Against the oracle, against the spectrum path on every sample, ragged calls bit for bit."""
    if any(G.defaults.get(k) for k in ("FDC_FORCE_GENERIC", "FDC_NO_POLY", "FDC_NO_BLOCK")):
        pytest.skip("suite run under a forced path")
    N = 65536
    H = N - N // R
    prm = [G.get_opt_channelparams(N, R, ((k + 0.5) / C) % 1.0, 1.0 / C) for k in range(C)]
    L = prm[0][1]
    assert L == 2 * N // C and sorted(set(p_[0] % L for p_ in prm[1:C - 1])) == [L // 4, 3 * L // 4]
    # the first and the last channel are clamped at the band edges by the derivation: dropped here (they sit on the grid: a third bank)
    chans = [(f, l, pb, sb) for (f, l, _lo, pb, sb) in prm[1:C - 1]]
    lout = L - L // R
    x = noise(nb * H, 9300 + C + R)
    p = G.Pipeline(N, R, chans, windowtype=1, max_blocks=nb)
    assert p.path() == 3 and "quarter" in p.describe() and "two launches" in p.describe()
    outs = p.work(x)
    n = len(chans)
    check = [0, 1, 2, 3, n // 2, n // 2 + 1, n - 2, n - 1]
    sub = [chans[c] for c in check]
    k = min(nb, 3)
    ref, _ = oracle.channelizer(N, R, 1, sub, x[:k * H], nthreads=8)
    t0 = nb - k
    ref2, _ = oracle.channelizer(N, R, 1, sub, x[t0 * H:], prefix=x[t0 * H - N // R:t0 * H], first_block=t0, nthreads=8)
    for i, c in enumerate(check):
        assert outs[c].size == nb * lout
        assert_close(outs[c][:k * lout], ref[i], "channel %d head" % c)
        assert_close(outs[c][t0 * lout:], ref2[i], "channel %d tail" % c)
    q = G.Pipeline(N, R, chans, windowtype=1, max_blocks=nb, flags=G.FDC_PIPE_NO_BLOCK)
    assert q.path() == 1
    for c, (a, b_) in enumerate(zip(outs, q.work(x))):
        assert_close(a, b_, "channel %d vs the spectrum path" % c)
    p.reset()
    cuts = [(0, 1), (1, 3), (3, max(3, nb // 2)), (max(3, nb // 2), nb)]
    parts = [p.work(x[a * H:b_ * H]) for a, b_ in cuts if b_ > a]
    for c in range(len(chans)):
        assert np.array_equal(np.concatenate([pp[c] for pp in parts]), outs[c])


@pytest.mark.parametrize("C,R,nb", [(1024, 2, 9), (2048, 4, 7)])
def test_the_whole_raster_of_half_overlapping_narrow_channels_is_three_banks(oracle, C, R, nb):
    """All C channels of the plan above: the first and the last slice are clamped at the band edges onto the grid — the same slice, N - l — beside the two
    banks a quarter and three quarters of a channel off it: three banks, three launches and one copy, where the cost rule still prefers that to the spectrum
    path.  The edge channels and their neighbours against the oracle, every sample against the spectrum path."""
    if any(G.defaults.get(k) for k in ("FDC_FORCE_GENERIC", "FDC_NO_POLY", "FDC_NO_BLOCK")):
        pytest.skip("suite run under a forced path")
    N = 65536
    H = N - N // R
    prm = [G.get_opt_channelparams(N, R, ((k + 0.5) / C) % 1.0, 1.0 / C) for k in range(C)]
    L = prm[0][1]
    assert prm[0][0] == prm[C - 1][0] == N - L
    chans = [(f, l, pb, sb) for (f, l, _lo, pb, sb) in prm]
    lout = L - L // R
    x = noise(nb * H, 9400 + C + R)
    p = G.Pipeline(N, R, chans, windowtype=1, max_blocks=nb)
    assert p.path() == 3 and "three launches" in p.describe() and "1 copies" in p.describe(), p.describe()
    outs = p.work(x)
    assert np.array_equal(outs[0], outs[C - 1])
    check = [0, 1, 2, C // 2, C - 2, C - 1]
    ref, _ = oracle.channelizer(N, R, 1, [chans[c] for c in check], x, nthreads=8)
    for i, c in enumerate(check):
        assert outs[c].size == nb * lout
        assert_close(outs[c], ref[i], "channel %d" % c)
    q = G.Pipeline(N, R, chans, windowtype=1, max_blocks=nb, flags=G.FDC_PIPE_NO_BLOCK)
    assert q.path() == 1
    for c, (a, b_) in enumerate(zip(outs, q.work(x))):
        assert_close(a, b_, "channel %d vs the spectrum path" % c)
    p.reset()
    parts = [p.work(x[a * H:b_ * H]) for a, b_ in [(0, 2), (2, 3), (3, nb)]]
    for c in range(len(chans)):
        assert np.array_equal(np.concatenate([pp[c] for pp in parts]), outs[c])
