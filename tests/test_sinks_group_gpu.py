"""GPU tests (-m gpu): the sink bank cut by frequency band over several members (fdc_sinks_group, SURVEY.md §8e "shard by channel /
by segment") against ONE bank on the same items.  The members are virtual (device 0 named several times): what is tested is the
cut of the bank into runs, the band every member copies (fdc_sinks_read_band / fdc_sinks_work_band: bins outside the band never
reach the member), the segment numbers in the IDs (seg_id_base), and the merge of the members' PDUs into one bank's emission
order.  Both sides run the same kernels on the same bins: number, order, every metadata field and every payload sample equal."""
import ctypes as C

import numpy as np
import pytest

import gr_fdc_amd as G
from gr_fdc_amd import _lib
from test_sinks_engines_gpu import onoff_spectrum, run_calls, same

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0], [0] * 7])
@pytest.mark.parametrize("maxblocks", [-1, 3])
def test_pac_bank_over_members_equals_one_bank(devices, maxblocks):
    N, R, nb = 2048, 2, 150
    rng = np.random.default_rng(100 + maxblocks)
    plan, carriers = [], []
    for c in range(70):
        cf = (c + 0.5) / 72 + 0.005
        bw = (0.004, 0.008, 0.002)[c % 3]
        plan.append((cf, bw, 100 + c))
        carriers.append((int(round((cf - bw / 2) * N)), int(round((cf + bw / 2) * N))))
    spec = onoff_spectrum(N, nb, carriers, 7 + maxblocks)
    cuts = sorted(set(int(v) for v in rng.integers(1, nb, 9)))
    one = G.Sinks(N, R, pac=plan, pac_thresh=6.0, pac_maxblocks=maxblocks, max_blocks=40)
    grp = G.SinksGroup(N, R, devices, pac=plan, pac_thresh=6.0, pac_maxblocks=maxblocks, max_blocks=40)
    mem = grp.members()
    assert len(mem) == len(devices) and sum(m[3] for m in mem) == len(plan)
    # bank order is frequency order here: every member copies a band of about 1 / members of the block
    assert all(0 <= lo < hi <= N for (_d, lo, hi, _p, _s) in mem)
    assert max(hi - lo for (_d, lo, hi, _p, _s) in mem) <= N // len(devices) + 64
    a, b = run_calls(grp, spec, cuts), run_calls(one, spec, cuts)
    assert len(b) > 200
    same(a, b, "pac bank over %d members, maxblocks %d" % (len(devices), maxblocks))


def test_unsorted_pac_bank_is_cut_by_frequency_and_emits_in_bank_order():
    """the group sorts the PowerActivationChannels by centre frequency before it cuts the bank (a member's band stays narrow whatever
    order cfg->pac[] lists them in) and puts the PDUs back into the bank's own emission order (fdc_sinks_pdu_emit_order)."""
    N, R, nb = 2048, 2, 120
    rng = np.random.default_rng(321)
    plan, carriers = [], []
    for c in range(60):
        cf = (c + 0.5) / 62 + 0.005
        bw = (0.004, 0.008, 0.002)[c % 3]
        plan.append((cf, bw, 500 + c))
        carriers.append((int(round((cf - bw / 2) * N)), int(round((cf + bw / 2) * N))))
    order = [int(v) for v in rng.permutation(len(plan))]
    shuffled = [plan[i] for i in order]
    spec = onoff_spectrum(N, nb, carriers, 99)
    cuts = sorted(set(int(v) for v in rng.integers(1, nb, 7)))
    one = G.Sinks(N, R, pac=shuffled, pac_thresh=6.0, pac_maxblocks=3, max_blocks=40)
    for devices in ([0, 0], [0, 0, 0, 0]):
        grp = G.SinksGroup(N, R, devices, pac=shuffled, pac_thresh=6.0, pac_maxblocks=3, max_blocks=40)
        mem = grp.members()
        assert max(hi - lo for (_d, lo, hi, _p, _s) in mem) <= N // len(devices) + 64        # bands, although the list is shuffled
        a, b = run_calls(grp, spec, cuts), run_calls(G.Sinks(N, R, pac=shuffled, pac_thresh=6.0, pac_maxblocks=3, max_blocks=40), spec, cuts)
        assert len(b) > 150
        same(a, b, "shuffled pac bank over %d members" % len(devices))
    del one


@pytest.mark.parametrize("variant", [0, 1])
def test_detection_segments_over_members_equal_one_bank(variant):
    N, R, nb = 4096, 2, 120
    segs = [(0.03, 0.22), (0.27, 0.47), (0.53, 0.72), (0.76, 0.97)]
    rng = np.random.default_rng(50 + variant)
    carriers, pos = [], 0.04
    while pos < 0.93:
        w = float(rng.uniform(0.004, 0.03))
        carriers.append((int(pos * N), int((pos + w) * N)))
        pos += w + float(rng.uniform(0.012, 0.05))
    spec = onoff_spectrum(N, nb, carriers, 21)
    cuts = sorted(set(int(v) for v in rng.integers(1, nb, 7)))
    kw = dict(segments=segs, det_thresh=10.0, det_maxblocks=3, minchandist=0.005, det_delay=1, puffer=0.2, max_blocks=48,
              det_variant=variant)
    one = G.Sinks(N, R, **kw)
    for devices in ([0, 0], [0, 0, 0], [0, 0, 0, 0, 0, 0]):        # six members, four segments: two members stay idle
        grp = G.SinksGroup(N, R, devices, **kw)
        assert sum(m[4] for m in grp.members()) == len(segs)
        one_pdus = run_calls(G.Sinks(N, R, **kw), spec, cuts)
        got = run_calls(grp, spec, cuts)
        assert len(one_pdus) > 20 and {m["source"] for (m, _d) in one_pdus} == {0, 1, 2, 3}
        same(got, one_pdus, "segments over %d members, variant %d" % (len(devices), variant))
    del one


def test_mixed_bank_bands_and_stale_bins():
    """PowerActivationChannels and segments in one bank.  The members must not depend on bins outside their band: the items are fed
    once as they are and once with everything outside each member's band replaced by large garbage — per member, through
    fdc_sinks_work_band — and the PDUs must not change."""
    N, R, nb = 4096, 2, 60
    plan = [((c + 0.5) / 40 * 0.45 + 0.02, 0.006, c) for c in range(20)]
    segs = [(0.55, 0.70), (0.75, 0.95)]
    carriers = [(int(round((cf - bw / 2) * N)), int(round((cf + bw / 2) * N))) for (cf, bw, _i) in plan]
    carriers += [(int(0.58 * N), int(0.60 * N)), (int(0.64 * N), int(0.66 * N)), (int(0.8 * N), int(0.83 * N)), (int(0.9 * N), int(0.91 * N))]
    spec = onoff_spectrum(N, nb, carriers, 77)
    kw = dict(pac=plan, pac_thresh=6.0, pac_maxblocks=4, segments=segs, det_thresh=10.0, det_maxblocks=3, minchandist=0.005,
              det_delay=1, puffer=0.2, max_blocks=32)
    one = G.Sinks(N, R, **kw)
    grp = G.SinksGroup(N, R, [0, 0], **kw)
    ref = run_calls(one, spec, [13, 32])
    same(run_calls(grp, spec, [13, 32]), ref, "mixed bank over two members")
    assert any(m["kind"] == 0 for (m, _d) in ref) and any(m["kind"] == 1 for (m, _d) in ref)
    # a single bank fed through its band only: garbage outside the band is never copied
    lo, hi = C.c_int32(), C.c_int32()
    pac_only = G.Sinks(N, R, pac=plan, pac_thresh=6.0, pac_maxblocks=4, max_blocks=64)
    _lib.check(_lib.lib().fdc_sinks_read_band(pac_only._h, C.byref(lo), C.byref(hi)))
    assert 0 < lo.value < hi.value < N // 2 + 64
    dirty = spec.copy()
    dirty[:, :lo.value] = 1e6
    dirty[:, hi.value:] = 1e6
    _lib.check(_lib.lib().fdc_sinks_work_band(pac_only._h, dirty.ctypes.data, nb, lo.value, hi.value))
    same(pac_only.pdus(), run_calls(G.Sinks(N, R, pac=plan, pac_thresh=6.0, pac_maxblocks=4, max_blocks=64), spec, []), "band-fed bank")
    with pytest.raises(G.FdcError):                                     # a band that does not cover what the bank reads is refused
        _lib.check(_lib.lib().fdc_sinks_work_band(pac_only._h, dirty.ctypes.data, 1, lo.value + 8, hi.value))
