"""Multi-device handle (fdc_pipeline_group, SURVEY.md §8e / VERDICT r03 row e2) on ONE GPU: the members are virtual — the
same device named two or three times — so what is tested is the dispatcher: the span cut, the halo of every span (the group's
history for span 0, the caller's buffer for the others), the global first-block index (window phase of odd f across span
boundaries), the output block offsets, the history and counter carried across calls.

The members run the kernels one handle would run on the same blocks (conftest: min_block_launch = 1, so launch length does not
change the kernel choice), and every block's arithmetic is independent of where it sits in a launch: the group's output must
equal the single handle's BIT FOR BIT."""
import ctypes as C

import numpy as np
import pytest

import gr_fdc_amd as G
from gr_fdc_amd import _lib

pytestmark = pytest.mark.gpu


def noise(n, seed):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)


def same_bits(a, b):
    return a.shape == b.shape and (a.view(np.uint32) == b.view(np.uint32)).all()


PLANS = {
    # the headline geometry, a subset of slots: one-kernel path (3)
    "uniform65536": (65536, 2, [(256 * c, 256, 0.88, 1.0) for c in (0, 1, 17, 128, 255)]),
    # odd offset: the window phase alternates with the GLOBAL block index, path 3 (OFF variant)
    "offset37": (65536, 2, [(256 * c + 37, 256, 0.88, 1.0) for c in (0, 5, 200, 254)]),
    # mixed widths, odd f: spectrum path (1); R = 4: four phase states
    "mixed65536": (65536, 4, [(1001, 256, 0.8, 1.0), (20000, 512, 0.7, 0.9), (40003, 1024, 0.6, 0.85), (60001, 128, 0.88, 1.0)]),
    # configs[0]-like: N = 4096, generic / register kernels (0)
    "cfg1": (4096, 2, [(2413, 256, 0.8, 1.0), (2901, 512, 0.82, 1.0), (1211, 512, 0.98, 1.0), (1793, 512, 0.66, 0.9)]),
    # R = 8, small block: many phase states, tiny halo
    "r8": (1024, 8, [(3, 64, 0.7, 0.9), (517, 128, 0.88, 1.0), (900, 32, 0.5, 0.8)]),
}


@pytest.mark.parametrize("name", sorted(PLANS))
@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
def test_group_equals_one_handle_bit_for_bit(name, devices):
    N, R, plan = PLANS[name]
    H = N - N // R
    calls = [24, 7, 1, 40, 3, 16] if N >= 65536 else [37, 5, 64, 1, 2, 19]          # ragged: spans of 0, 1 and many blocks
    total = sum(calls)
    x = noise(total * H, 99 + N + R)
    one = G.Pipeline(N, R, plan, windowtype=1, max_blocks=max(calls))
    grp = G.PipelineGroup(N, R, plan, devices, windowtype=1, max_blocks=max(calls), min_span_blocks=2)
    assert grp.size() == len(devices) and grp.path() == one.path()
    b = 0
    for n in calls:
        seg = x[b * H:(b + n) * H]
        ref = one.work(seg)
        got = grp.work(seg)
        spans = grp.last_spans()
        assert sum(k for (_f, k) in spans) == n
        used = [s for s in spans if s[1] > 0]
        assert used[0][0] == b and all(used[i][0] + used[i][1] == used[i + 1][0] for i in range(len(used) - 1))
        assert len(used) == max(1, min(len(devices), n // 2))
        for c, (r_, g_) in enumerate(zip(ref, got)):
            assert same_bits(r_, g_), "%s devices=%s call of %d blocks at %d, channel %d" % (name, devices, n, b, c)
        b += n
    one.close(); grp.close()


def test_group_against_the_oracle(oracle):
    """The dispatcher against the CPU restatement directly (not only against the product's own single handle)."""
    N, R, plan = PLANS["cfg1"]
    H = N - N // R
    x = noise(48 * H, 5)
    grp = G.PipelineGroup(N, R, plan, [0, 0, 0], windowtype=1, max_blocks=32, min_span_blocks=1)
    got = [np.concatenate(p) for p in zip(grp.work(x[:31 * H]), grp.work(x[31 * H:]))]
    ref, _ = oracle.channelizer(N, R, 1, plan, x)
    for g_, r_ in zip(got, ref):
        err = np.linalg.norm(g_ - r_) / np.linalg.norm(r_)
        assert err <= 1e-5 and np.abs(g_ - r_).max() <= 1e-5 * np.abs(r_).max()


def test_group_spectrum_port_and_real_input():
    N, R = 4096, 2
    plan = [(2413, 256, 0.8, 1.0), (100, 512, 0.7, 0.9)]
    H = N - N // R
    x = noise(20 * H, 11)
    one = G.Pipeline(N, R, plan, max_blocks=20, keep_spectrum=True)
    grp = G.PipelineGroup(N, R, plan, [0, 0], max_blocks=20, min_span_blocks=1, keep_spectrum=True)
    for lo, hi in [(0, 9), (9, 20)]:
        (ro, rs), (go, gs) = one.work(x[lo * H:hi * H], want_spectrum=True), grp.work(x[lo * H:hi * H], want_spectrum=True)
        assert same_bits(rs, gs)
        assert all(same_bits(a, b) for a, b in zip(ro, go))
    one.close(); grp.close()
    # float input: the halo of a span is float too
    xr = np.random.default_rng(3).standard_normal(20 * H).astype(np.float32)
    one = G.Pipeline(N, R, plan, max_blocks=20)
    grp = G.PipelineGroup(N, R, plan, [0, 0, 0], max_blocks=20, min_span_blocks=1)
    for lo, hi in [(0, 13), (13, 20)]:
        ro, go = one.work_real(xr[lo * H:hi * H]), grp.work_real(xr[lo * H:hi * H])
        assert all(same_bits(a, b) for a, b in zip(ro, go))
    # mixing the two item types on one stream is refused, reset makes the group a fresh stream
    with pytest.raises(G.FdcError):
        grp.work(x[:H])
    grp.reset(); one.reset()
    assert all(same_bits(a, b) for a, b in zip(one.work(x[:5 * H]), grp.work(x[:5 * H])))


def test_group_with_pinned_buffers_and_errors():
    """Registered (pinned) caller buffers: every member DMAs its span in place and stores straight into the caller's per-channel
    buffers at the span's offset."""
    N, R, plan = PLANS["uniform65536"]
    H = N - N // R
    nb = 36
    x = np.empty(nb * H, dtype=np.complex64)
    x[:] = noise(nb * H, 21)
    one = G.Pipeline(N, R, plan, max_blocks=nb)
    ref = one.work(x)
    grp = G.PipelineGroup(N, R, plan, [0, 0, 0], max_blocks=nb, min_span_blocks=4)
    outs = [np.zeros(nb * lo, dtype=np.complex64) for lo in grp.lout]
    G.register_host(x)
    for o in outs:
        G.register_host(o)
    try:
        grp.work(x, outs=outs)
        assert all(same_bits(a, b) for a, b in zip(ref, outs))
    finally:
        G.unregister_host(x)
        for o in outs:
            G.unregister_host(o)
    with pytest.raises(G.FdcError):
        grp.work(np.zeros((nb + 1) * H, dtype=np.complex64))           # above max_blocks
    with pytest.raises((G.FdcError, ValueError)):
        G.PipelineGroup(N, R, plan, [0, 99], max_blocks=8)              # no such device
    with pytest.raises((G.FdcError, ValueError)):
        G.PipelineGroup(N, R, plan, [], max_blocks=8)
    # a member's capacity covers the longest span the policy can give it
    g2 = G.PipelineGroup(4096, 2, PLANS["cfg1"][2], [0, 0, 0, 0], max_blocks=100, min_span_blocks=16)
    assert g2.member_max_blocks() == 31                                   # 31 blocks: one member (31 // 16 == 1)
    g2.work(noise(31 * 2048, 1)); assert [n for (_f, n) in g2.last_spans()] == [31, 0, 0, 0]
    g2.work(noise(100 * 2048, 2)); assert [n for (_f, n) in g2.last_spans()] == [25, 25, 25, 25]


def test_hier_block_on_a_group():
    args = dict(inptype=8, inpveclen=1, blocksize=4096, relinvovl=2,
                throughput_channels=[(0.12, 0.05), (0.22, 0.1), (-0.14, 0.12)], activity_controlled_channels=[],
                act_contr_threshold=6.0, fs=1.0, centerfrequency=0.0, freqmode=0, windowtype=1, msgoutput=False,
                fileoutput=False, outputpath=".", threaded=False, activity_detection_segments=[], act_det_threshold=6.0,
                minchandist=0.01, act_det_deactivation_delay=1, minchanflankpuffer=0.2, verbose=0, pow_act_deactivation_delay=1,
                pow_act_maxblocks=8, act_det_maxblocks=8, debug=False)
    a = G.FrequencyDomainChannelizer(**args, max_blocks=16)
    b = G.FrequencyDomainChannelizer(**args, max_blocks=16, devices=[0, 0])
    assert isinstance(b.pipeline, G.PipelineGroup)
    x = noise(16 * 2048, 8)
    for lo, hi in [(0, 16 * 2048), (0, 5 * 2048)]:
        assert all(same_bits(p, q) for p, q in zip(a.work(x[lo:hi]), b.work(x[lo:hi])))


def test_selftest_runs_the_group_over_all_devices():
    assert _lib.check(_lib.lib().fdc_selftest_devices()) == _lib.lib().fdc_device_count()
