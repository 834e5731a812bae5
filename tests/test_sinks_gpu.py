"""GPU tests (-m gpu) of the stateful sinks against the oracle restatements: PDU metadata exact (ints) / 1e-12
(doubles), emitted-sample counts exact, payload within 1e-5 relative (SURVEY.md §8d parity metric).
Bursts sit >= 20 dB above the floor so threshold decisions do not depend on the float summation order."""
import json
import os

import numpy as np
import pytest

import gr_fdc_amd as G

pytestmark = pytest.mark.gpu
TOL = 1e-5
INTS = ("kind", "source", "chan_id", "finalized", "part", "has_part", "blockstart", "blockend")


def unstamp(ident):
    """ID without its activation timestamp ("YYYY-mm-dd-HH-MM-SS." — non-deterministic); the prefix
    itself must have the reference's strftime shape."""
    import re
    assert re.match(r"^\d{4}-\d{2}-\d{2}-\d{2}-\d{2}-\d{2}\.", ident), ident
    return ident[20:]


def burst_spectrum(N, nb, bursts, seed, floor=1e-3):
    """normalised-spectrum items: white floor plus rectangular bursts (lo_bin, hi_bin, first_block, last_block, amp)"""
    rng = np.random.default_rng(seed)
    s = floor * (rng.standard_normal((nb, N)) + 1j * rng.standard_normal((nb, N)))
    for lo, hi, b0, b1, amp in bursts:
        s[b0:b1 + 1, lo:hi] += amp * (rng.standard_normal((b1 - b0 + 1, hi - lo)) + 1j * rng.standard_normal((b1 - b0 + 1, hi - lo)))
    return s.astype(np.complex64)


def compare(got, ref, vec=True):
    assert len(got) == len(ref), (len(got), len(ref))
    for (gm, gd), r in zip(got, ref):
        for k in INTS:
            assert int(gm[k]) == int(r[k]), (k, gm, {q: r[q] for q in r if q != "samples"})
        if vec:
            assert gm["vectorstart"] == r["vectorstart"] and gm["vectorend"] == r["vectorend"]
        assert abs(gm["rel_bw"] - r["rel_bw"]) < 1e-12 and abs(gm["rel_cfreq"] - r["rel_cfreq"]) < 1e-12
        assert gd.size == r["samples"].size
        if gd.size:
            d = gd.astype(np.complex128) - r["samples"].astype(np.complex128)
            assert np.linalg.norm(d) <= TOL * np.linalg.norm(r["samples"])
            assert np.abs(d).max() <= TOL * np.abs(r["samples"]).max()


def test_pac_known_answer_from_survey(oracle, golden_dir):
    """The scenario of SURVEY.md §8c: bins 1600-1799 active in blocks 3-7, N=4096, R=4 -> one finalised PDU."""
    ka = json.load(open(os.path.join(golden_dir, "sink_known_answers.json")))["PowerActivationChannel"]
    N, R = 4096, 4
    spec = burst_spectrum(N, 12, [(1600, 1800, 3, 7, 1.0)], 0)
    cf, bw = (1600 + 1800) / 2 / N, 200 / N
    bank = G.Sinks(N, R, pac=[(cf, bw, 0)], pac_thresh=6.0, pac_maxblocks=-1, max_blocks=16)
    got = bank.work(spec)
    assert len(got) == 1
    m, d = got[0]
    assert m["finalized"] and abs(m["rel_bw"] - ka["rel_bw"]) < 1e-12
    assert (m["blockstart"], m["blockend"], d.size) == (ka["blockstart"], ka["blockend"], ka["nsamples"])
    compare(got, oracle.PowerActivationChannel(N, cf, bw, R, 6.0, -1, 0, 0).work(spec), vec=False)


@pytest.mark.parametrize("maxblocks", [-1, 0, 3])
def test_pac_bank_vs_oracle(oracle, maxblocks):
    N, R, nb = 4096, 2, 40
    plan = [(0.20, 0.03, 0), (0.41, 0.05, 1), (0.70, 0.011, 2), (0.9, 0.1, 7)]
    bursts = []
    rng = np.random.default_rng(3)
    for cf, bw, _ in plan:
        lo, hi = int(round((cf - bw / 2) * N)), int(round((cf + bw / 2) * N))
        t = 2
        while t < nb - 3:
            ln = int(rng.integers(2, 9))
            bursts.append((lo, hi, t, min(nb - 2, t + ln), 1.0))
            t += ln + int(rng.integers(3, 7))
    spec = burst_spectrum(N, nb, bursts, 11)
    bank = G.Sinks(N, R, pac=plan, pac_thresh=6.0, pac_maxblocks=maxblocks, max_blocks=16)
    got = bank.work(spec[:7].reshape(-1)) + bank.work(spec[7:].reshape(-1))       # state carries over calls and batches
    assert len(got) > 4
    for i, (cf, bw, ident) in enumerate(plan):
        o = oracle.PowerActivationChannel(N, cf, bw, R, 6.0, maxblocks, 0, ident)
        p = bank.pac_params(i)
        assert (p["extract_start"], p["extract_stop"], p["extract_width"], p["measure_start"], p["measure_stop"],
                p["output_len"]) == (o.extract_start, o.extract_stop, o.extract_width, o.measure_start, o.measure_stop, o.output_len)
        ref = o.work(spec[:7]) + o.work(spec[7:])
        compare([g for g in got if g[0]["source"] == ident], ref, vec=False)


def test_pac_face_and_errors(oracle, tmp_path):
    N, R = 1024, 2
    spec = burst_spectrum(N, 10, [(300, 340, 2, 5, 1.0)], 5)
    blk = G.PowerActivationChannel(N, 320 / N, 40 / N, R, 6.0, -1, 0, True, True, str(tmp_path), 0, 3)
    pdus = blk.work(spec)
    assert len(pdus) == 1 and unstamp(pdus[0][0]["ID"]) == "PowActChan.3.0.fin" and pdus[0][0]["finalized"] is True
    assert set(pdus[0][0]) == {"ID", "finalized", "part", "rel_cfreq", "rel_bw", "blockstart", "blockend"}
    f = np.fromfile(os.path.join(str(tmp_path), pdus[0][0]["ID"]), dtype=np.complex64)
    assert (f == pdus[0][1]).all()
    with pytest.raises(ValueError):
        G.PowerActivationChannel(N, 0.01, 0.1, R, 6.0, -1, 0, False, False, "", 0, 0)    # out of band (…cc:318-319)
    with pytest.raises(ValueError):
        G.PowerActivationChannel(N, 0.5, 0.1, R, 0.0, -1, 0, False, False, "", 0, 0)     # thresh <= 0 (:378-379)
    with pytest.raises(ValueError):
        G.PowerActivationChannel(N, 0.5, 0.1, 3, 6.0, -1, 0, False, False, "", 0, 0)     # relinvovl not 2^k (:68-69)


@pytest.mark.parametrize("N,R,maxblocks,delay", [(4096, 4, -1, 1), (4096, 2, 0, 0), (16384, 2, 3, 2), (65536, 2, 128, 1)])
def test_vcm_vs_oracle(oracle, N, R, maxblocks, delay):
    nb = 36
    segs = [[0.05, 0.45], [0.55, 0.95]]
    rng = np.random.default_rng(N + R)
    bursts = []
    for s0, s1 in segs:
        pos = s0 + 0.02
        while pos < s1 - 0.06:
            wdt = float(rng.uniform(0.004, 0.03))
            lo, hi = int(pos * N), int((pos + wdt) * N)
            t0 = int(rng.integers(1, 12)); ln = int(rng.integers(4, 16))
            bursts.append((lo, hi, t0, min(nb - 3, t0 + ln), 1.0))
            pos += wdt + float(rng.uniform(0.03, 0.06))
    spec = burst_spectrum(N, nb, bursts, 17)
    blk = G.Sinks(N, R, segments=[tuple(s) for s in segs], det_thresh=10.0, det_maxblocks=maxblocks, minchandist=0.005,
                  det_delay=delay, puffer=0.2, max_blocks=16)
    o = oracle.ActivityDetectionVcm(N, segs, 10.0, R, maxblocks, 0.005, delay, 0.2)
    for i, g in enumerate(o.segments):
        assert blk.segment_params(i) == g
    got = blk.work(spec[:5].reshape(-1)) + blk.work(spec[5:].reshape(-1))
    ref = o.work(spec[:5]) + o.work(spec[5:])
    assert len(ref) >= 4
    compare(got, ref)


def test_vcm_known_answer_and_face(oracle, golden_dir):
    ka = json.load(open(os.path.join(golden_dir, "sink_known_answers.json")))["activity_detection_channelizer_vcm"]
    N, R = 4096, 4
    spec = burst_spectrum(N, 12, [(1600, 1800, 3, 7, 1.0)], 0)
    blk = G.activity_detection_channelizer_vcm(N, [[0.3, 0.55]], 10.0, R, -1, True, False, "", False, 0.005, 1, 0.2, 0)
    pdus = blk.work(spec)
    assert len(pdus) == 1
    d, data = pdus[0]
    assert unstamp(d["ID"]) == "DETECTED.0.0" and d["finalized"] is True and "part" not in d
    assert (abs(d["rel_bw"] - ka["rel_bw"]) < 1e-12 and d["blockstart"] == ka["blockstart"] and d["blockend"] == ka["blockend"]
            and data.size == ka["nsamples"])
    # the survey run's segment geometry is not recorded: the detection grid (dec = 10 bins) may shift the slice
    assert abs(d["vectorstart"] - ka["vectorstart"]) <= 10 and d["vectorend"] - d["vectorstart"] == 512
    with pytest.raises(ValueError):
        G.activity_detection_channelizer_vcm(N, [[0.5, 0.3]], 10.0, R, -1, True, False, "", False, 0.005, 1, 0.2, 0)
    with pytest.raises(ValueError):
        G.activity_detection_channelizer_vcm(N, [[0.1, 0.3]], 10.0, R, -1, True, False, "", False, 1.5, 1, 0.2, 0)


def test_hier_block_with_sinks_from_device_spectrum(oracle):
    """FrequencyDomainChannelizer mirror with throughput channels + activity-controlled channels + detection segments:
    the sinks are fed from the device-resident spectrum of the same call (fdc_pipeline_work_sinks)."""
    N, R, nb = 4096, 4, 24
    H = N - N // R
    rng = np.random.default_rng(8)
    n = np.arange(nb * H)
    x = 0.01 * (rng.standard_normal(nb * H) + 1j * rng.standard_normal(nb * H))
    # a bursty carrier inside the activity-controlled channel and one inside the detection segment
    for fc, t0, t1 in [(-0.2, 5, 11), (0.31, 8, 17)]:
        env = np.zeros(nb * H); env[t0 * H:t1 * H] = 1.0
        sym = (rng.integers(0, 2, nb * H // 64 + 1) * 2 - 1) + 1j * (rng.integers(0, 2, nb * H // 64 + 1) * 2 - 1)
        x += env * np.repeat(sym, 64)[:nb * H] * np.exp(2j * np.pi * fc * n)
    x = x.astype(np.complex64)
    fdc = G.FrequencyDomainChannelizer(8, 1, N, R, [[0.1, 0.05]], [[-0.2, 0.04]], 6.0, 1.0, 0.0, 'normalized', 1,
                                       True, False, "", False, [[0.25, 0.4]], 10.0, 0.005, 1, 0.2, 0, 0, -1, -1, True,
                                       max_blocks=nb)
    ports = fdc.work(x)
    spec = ports[0]
    pac_ref = oracle.PowerActivationChannel(N, (-0.2 + 0.5) % 1.0, 0.04, R, 6.0, -1, 0, 0).work(spec)
    det_ref = oracle.SegmentDetection(0, N, R, 0.75, 0.9, 10.0, 0.005, 0.2, -1, 1).work(spec)
    got_pac = [(d, s) for (d, s) in fdc.messages if unstamp(d["ID"]).startswith("PowActChan")]
    got_det = [(d, s) for (d, s) in fdc.messages if unstamp(d["ID"]).startswith("DETECTED")]
    assert len(pac_ref) >= 1 and len(det_ref) >= 1
    assert len(got_pac) == len(pac_ref) and len(got_det) == len(det_ref)
    for (d, s), r in zip(got_pac + got_det, pac_ref + det_ref):
        assert (d["blockstart"], d["blockend"], s.size) == (r["blockstart"], r["blockend"], r["samples"].size)
        assert np.abs(s - r["samples"]).max() <= 1e-5 * np.abs(r["samples"]).max()


@pytest.mark.parametrize("N,R,maxblocks,delay", [(4096, 4, -1, 1), (4096, 2, 0, 0), (16384, 2, 3, 2)])
def test_segment_detection_face_vs_oracle(oracle, golden_dir, N, R, maxblocks, delay):
    """SegmentDetection (the twin the hier block uses): own geometry, raw sums, counter from 0, separate partial pass."""
    nb = 36
    rng = np.random.default_rng(N + R + 5)
    bursts = []
    pos = 0.12
    while pos < 0.78:
        wdt = float(rng.uniform(0.004, 0.03))
        t0 = int(rng.integers(1, 12)); ln = int(rng.integers(4, 16))
        bursts.append((int(pos * N), int((pos + wdt) * N), t0, min(nb - 3, t0 + ln), 1.0))
        pos += wdt + float(rng.uniform(0.03, 0.06))
    spec = burst_spectrum(N, nb, bursts, 23)
    blk = G.SegmentDetection(3, N, R, 0.1, 0.8, 10.0, 0.005, 0.2, maxblocks, delay, True, False, "", False, 0, max_blocks=16)
    o = oracle.SegmentDetection(3, N, R, 0.1, 0.8, 10.0, 0.005, 0.2, maxblocks, delay)
    assert blk.segment == o.segments[0]
    got = blk.work(spec[:5].reshape(-1)) + blk.work(spec[5:].reshape(-1))
    ref = o.work(spec[:5]) + o.work(spec[5:])
    assert len(ref) >= 3 and len(got) == len(ref)
    for (d, s), r in zip(got, ref):
        assert unstamp(d["ID"]) == "DETECTED.3.%d" % r["chan_id"]
        assert (d["finalized"], d["blockstart"], d["blockend"], d["vectorstart"], d["vectorend"]) == \
            (r["finalized"], r["blockstart"], r["blockend"], r["vectorstart"], r["vectorend"])
        assert ("part" in d) == r["has_part"] and s.size == r["samples"].size
        if s.size:
            assert np.abs(s - r["samples"]).max() <= TOL * np.abs(r["samples"]).max()
    ka = json.load(open(os.path.join(golden_dir, "sink_known_answers.json")))["SegmentDetection"]
    spec2 = burst_spectrum(4096, 12, [(1600, 1800, 3, 7, 1.0)], 0)
    (d, s), = G.SegmentDetection(0, 4096, 4, 0.3, 0.55, 10.0, 0.005, 0.2, -1, 1, True, False, "", False, 0).work(spec2)
    assert (d["blockstart"], d["blockend"], s.size) == (ka["blockstart"], ka["blockend"], ka["nsamples"]) and abs(d["rel_bw"] - ka["rel_bw"]) < 1e-12


def test_sinks_wider_than_one_workgroup_transform(oracle):
    """Extraction widths above 8192 bins (a PowerActivationChannel of 0.2 of a 65536-bin band -> 16384; a detected carrier
    of ~11000 bins -> 16384) run task by task through the two-pass inverse transform; same PDUs as the oracle."""
    N, R, nb = 65536, 2, 12
    cf, bw = 0.30, 0.20
    lo, hi = int(round((cf - bw / 2) * N)), int(round((cf + bw / 2) * N))
    spec = burst_spectrum(N, nb, [(lo, hi, 2, 6, 1.0), (40000, 51000, 4, 9, 1.0)], 23)
    bank = G.Sinks(N, R, pac=[(cf, bw, 4)], pac_thresh=6.0, pac_maxblocks=-1, segments=[(0.55, 0.95)], det_thresh=10.0,
                   det_maxblocks=-1, minchandist=0.005, det_delay=1, puffer=0.2, max_blocks=16)
    assert bank.pac_params(0)["extract_width"] == 16384
    got = bank.work(spec.reshape(-1))
    rp = oracle.PowerActivationChannel(N, cf, bw, R, 6.0, -1, 0, 4).work(spec)
    rd = oracle.ActivityDetectionVcm(N, [[0.55, 0.95]], 10.0, R, -1, 0.005, 1, 0.2).work(spec)
    assert len(rp) >= 1 and len(rd) >= 1 and max(r["samples"].size for r in rd) >= 8192 * 3
    compare([g for g in got if g[0]["kind"] == rp[0]["kind"]], rp, vec=False)
    compare([g for g in got if g[0]["kind"] != rp[0]["kind"]], rd)


def test_sinks_every_width_class_in_one_call(oracle):
    """PowerActivationChannels of extraction widths 128 ... 32768 in one bank and one call: the 256 kernel, the one launch for the
    other classes up to 4096 points, the task-addressed two-pass transform above that (8192, 16384, 32768) — every PDU against the
    oracle, on both engines."""
    N, R, nb = 65536, 2, 10
    widths = [128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768]
    plan, bursts, pos = [], [], 0.004
    for k, w in enumerate(widths):
        bw = 0.8 * w / N
        cf = pos + bw / 2
        plan.append((cf, bw, 10 + k))
        bursts.append((int(round((cf - bw / 2) * N)), int(round((cf + bw / 2) * N)), 1 + k % 3, 5 + k % 4, 1.0))
        pos += bw + 0.02
    assert pos < 1.0
    spec = burst_spectrum(N, nb, bursts, 77)
    # (look-ahead banks run the classes above 4096 points side by side on streams of their own: the same PDUs)
    for host, la in ((False, False), (True, False), (False, True)):
        bank = G.Sinks(N, R, pac=plan, pac_thresh=6.0, pac_maxblocks=3, max_blocks=16, host_decisions=host, lookahead=la)
        assert [bank.pac_params(i)["extract_width"] for i in range(len(widths))] == widths
        got = bank.work(spec[:4].reshape(-1)) + bank.work(spec[4:].reshape(-1)) if la else bank.work(spec.reshape(-1))
        ref = []
        for (cf, bw, ident) in plan:
            o = oracle.PowerActivationChannel(N, cf, bw, R, 6.0, 3, 0, ident)
            ref.append(o.work(spec[:4]) + o.work(spec[4:]) if la else o.work(spec))
        assert all(len(r) >= 1 for r in ref)
        for k, (cf, bw, ident) in enumerate(plan):
            mine = [g for g in got if g[0]["source"] == ident]
            compare(mine, ref[k], vec=False)


def test_pipeline_refuses_sinks_of_other_size():
    """fdc_pipeline_work_sinks writes the spectrum into the sinks' device buffer: a bank made for fewer blocks per call or
    another block length is refused before anything is written."""
    N, R = 1024, 2
    H = N - N // R
    p = G.Pipeline(N, R, [], max_blocks=8, keep_spectrum=True)
    small = G.Sinks(N, R, pac=[(0.5, 0.05, 0)], pac_thresh=6.0, pac_maxblocks=-1, max_blocks=4)
    other = G.Sinks(2 * N, R, pac=[(0.5, 0.05, 0)], pac_thresh=6.0, pac_maxblocks=-1, max_blocks=8)
    x = np.zeros(8 * H, np.complex64)
    for bank in (small, other):
        with pytest.raises(G.FdcError):
            p.work(x, sinks=bank)
    p.work(x[:4 * H], sinks=small)          # within capacity: fine


def test_message_ids_log_files_and_output_files(oracle, tmp_path, monkeypatch):
    """f2: ID = <activation time>.PowActChan.<ID>.<n> / <activation time>.DETECTED.<seg>.<n> (PowerActivationChannel_impl.cc:
    308-312, …vcm_impl.cc:526-530), the same string on every part of one activation; files <path>/<ID>.fin and
    <path>/<ID>.parted.<k> (:235-244, …vcm_impl.cc:431-439, :488-496); verbose = 2 writes the reference's log files
    (PowerActivationChannel_impl.cc:54, …vcm_impl.cc:94) with its lines."""
    monkeypatch.chdir(tmp_path)
    N, R, nb = 1024, 4, 16
    spec = burst_spectrum(N, nb, [(300, 340, 3, 12, 1.0)], 5)
    pac = G.PowerActivationChannel(N, 320.0 / N, 40.0 / N, R, 6.0, 3, 0, True, True, str(tmp_path), 2, 7)
    pdus = pac.work(spec)
    assert len(pdus) >= 3
    ids = [d["ID"] for d, _s in pdus]
    base = ids[0].rsplit(".", 1)[0]
    assert unstamp(base) == "PowActChan.7.0" and all(i.rsplit(".", 1)[0] == base for i in ids)      # one activation, one timestamp
    assert [i.rsplit(".", 1)[1] for i in ids[:-1]] == ["part"] * (len(ids) - 1) and ids[-1].endswith(".fin")
    for k, (d, s) in enumerate(pdus):
        fn = base + (".fin" if d["finalized"] else ".parted.%d" % d["part"])
        assert np.array_equal(np.fromfile(tmp_path / fn, dtype=np.complex64), s) and d["part"] == k
    log = open(tmp_path / "gr-FDC.PowActChan.7.log").read()
    assert log.startswith("\n############################") and "# extract_start: " in log and "# equivalent bw: " in log
    lines = [ln for ln in log.split("\n") if ln.startswith(base)]
    assert len(lines) == len(pdus) and lines[-1].startswith(base + ".fin: start=") and ", blockend=" in lines[-1]
    assert lines[0].startswith(base + ".parted.0: start=")
    # vcm face: one log file for the block, segment lines at construction, one line per emitted channel
    det = G.activity_detection_channelizer_vcm(N, [[0.2, 0.6]], 10.0, R, -1, True, True, str(tmp_path), False, 0.01, 1, 0.2, 2)
    dp = det.work(spec)
    assert len(dp) == 1 and unstamp(dp[0][0]["ID"]) == "DETECTED.0.0"
    assert np.array_equal(np.fromfile(tmp_path / (dp[0][0]["ID"] + ".fin"), dtype=np.complex64), dp[0][1])
    dlog = open(tmp_path / "gr-FDC.ActDetChan.log").read()
    assert "# Segment 0: \n# start: " in dlog and "# chan_decimation_fact: " in dlog and dp[0][0]["ID"] + ".fin: start=" in dlog
    # verbose = 1 prints, verbose = 0 stays silent and writes no log
    os.remove(tmp_path / "gr-FDC.PowActChan.7.log")
    G.PowerActivationChannel(N, 320.0 / N, 40.0 / N, R, 6.0, 3, 0, True, False, "", 0, 7).work(spec)
    assert not os.path.exists(tmp_path / "gr-FDC.PowActChan.7.log")


class _HipCopy:
    """Host-to-device copies on a given stream without torch (the test process has the library's HIP runtime loaded)."""
    def __init__(self):
        import ctypes as C
        self.C = C
        self.h = C.CDLL("/opt/rocm/lib/libamdhip64.so")
        self.h.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        self.h.hipStreamSynchronize.argtypes = [C.c_void_p]

    def upload(self, dst, arr, stream):
        assert self.h.hipMemcpyAsync(dst, arr.ctypes.data, arr.nbytes, 1, stream) == 0
        assert self.h.hipStreamSynchronize(stream) == 0      # (the source is pageable numpy memory: keep it alive until the copy is done)


@pytest.mark.parametrize("kind,host_decisions", [("pac", False), ("vcm", False), ("vcm", True), ("pac", True)])
def test_lookahead_bank_emits_the_same_pdus(kind, host_decisions):
    """FDC_SINKS_LOOKAHEAD (round 5): two spectrum buffers, the producer fills batch n + 1 on the fill stream (with or without its power
    cells) before batch n is submitted.  Same PDUs, metadata and payload bit for bit, as the one-buffer bank fed batch by batch: ragged
    batches, bursts that cross the batch boundaries (history block, buffered blocks, live channels), both engines."""
    N, R = 4096, 2
    sizes = [13, 16, 5, 16, 1, 16, 9]
    nb = sum(sizes)
    rng = np.random.default_rng(5)
    bursts = []
    if kind == "pac":
        plan = [(0.20, 0.03, 0), (0.41, 0.05, 1), (0.70, 0.011, 2), (0.9, 0.1, 7)]
        for cf, bw, _ in plan:
            lo, hi = int(round((cf - bw / 2) * N)), int(round((cf + bw / 2) * N))
            t = 2
            while t < nb - 3:
                ln = int(rng.integers(2, 14))
                bursts.append((lo, hi, t, min(nb - 2, t + ln), 1.0))
                t += ln + int(rng.integers(3, 7))
        kw = dict(pac=plan, pac_thresh=6.0, pac_maxblocks=3)
    else:
        segs = [(0.05, 0.45), (0.55, 0.95)]
        for s0, s1 in segs:
            pos = s0 + 0.02
            while pos < s1 - 0.06:
                wdt = float(rng.uniform(0.004, 0.03))
                t = int(rng.integers(1, 12))
                while t < nb - 3:
                    ln = int(rng.integers(4, 20))
                    bursts.append((int(pos * N), int((pos + wdt) * N), t, min(nb - 3, t + ln), 1.0))
                    t += ln + int(rng.integers(3, 9))
                pos += wdt + float(rng.uniform(0.03, 0.06))
        kw = dict(segments=segs, det_thresh=10.0, det_maxblocks=3, minchandist=0.005, det_delay=1, puffer=0.2)
    spec = burst_spectrum(N, nb, bursts, 23)
    cuts = np.cumsum([0] + sizes)
    batches = [np.ascontiguousarray(spec[cuts[i]:cuts[i + 1]].reshape(-1)) for i in range(len(sizes))]

    plain = G.Sinks(N, R, max_blocks=16, host_decisions=host_decisions, **kw)
    assert plain.spectrum_ahead_ptr() is None and plain.fill_stream() is None
    with pytest.raises(G.FdcError):
        plain.prepare(4)
    ref = []
    for b in batches:
        ref += plain.work(b)
    assert len(ref) >= 8

    hip = _HipCopy()
    bank = G.Sinks(N, R, max_blocks=16, host_decisions=host_decisions, lookahead=True, **kw)
    fs = bank.fill_stream()
    assert fs and bank.spectrum_ahead_ptr() and bank.spectrum_ahead_ptr() != bank.spectrum_ptr()
    hip.upload(bank.spectrum_ptr(), batches[0], fs)
    got = []
    for i in range(len(batches)):
        cur = bank.spectrum_ptr()
        if i + 1 < len(batches):
            hip.upload(bank.spectrum_ahead_ptr(), batches[i + 1], fs)
            if i % 2 == 0:                                     # every other batch: its power cells too, behind the fill
                bank.prepare(sizes[i + 1], ahead=True)
        got += bank.submit_device(sizes[i]) if bank.engine() == 1 else bank.work_device(sizes[i])
        assert bank.spectrum_ptr() != cur                      # the buffers have swapped
    got += bank.flush()
    assert len(got) == len(ref)
    for (gm, gd), (rm, rd) in zip(got, ref):
        assert {k: gm[k] for k in gm if k != "id"} == {k: rm[k] for k in rm if k != "id"}
        assert unstamp(gm["id"]) == unstamp(rm["id"])
        assert np.array_equal(gd, rd)


def _bursty_stream(N, R, nb, seed, carriers):
    """nb items of (N - N/R) samples: noise floor plus QPSK-like bursty carriers (fc, first_block, last_block)"""
    H = N - N // R
    rng = np.random.default_rng(seed)
    n = np.arange(nb * H)
    x = 0.01 * (rng.standard_normal(nb * H) + 1j * rng.standard_normal(nb * H))
    for fc, t0, t1 in carriers:
        env = np.zeros(nb * H); env[t0 * H:t1 * H] = 1.0
        sym = (rng.integers(0, 2, nb * H // 64 + 1) * 2 - 1) + 1j * (rng.integers(0, 2, nb * H // 64 + 1) * 2 - 1)
        x += env * np.repeat(sym, 64)[:nb * H] * np.exp(2j * np.pi * fc * n)
    return x.astype(np.complex64)


def _same_messages(got, ref):
    assert len(got) == len(ref), (len(got), len(ref))
    for (gd, gs), (rd, rs) in zip(got, ref):
        assert {k: gd[k] for k in gd if k != "ID"} == {k: rd[k] for k in rd if k != "ID"}
        assert unstamp(gd["ID"]) == unstamp(rd["ID"])
        assert np.array_equal(gs, rs)


@pytest.mark.parametrize("N,R,sizes,verbose", [(4096, 4, [5, 8, 1, 8, 2, 7, 8, 3], 0), (65536, 2, [24, 7, 24, 1, 16], 0), (4096, 2, [6, 6, 6, 6], 1)])
def test_hier_block_pipelined_emits_the_same_pdus(N, R, sizes, verbose, capfd):
    """fdc_pipeline_work_sinks on a look-ahead bank (round 6: the pipelined hier block — copy and forward transform of call n beside the
    sinks of call n - 1, PDUs handed out one or two calls later, fdc_pipeline_flush_sinks at the end) against the serial form: the same
    messages in the same order, payloads bit for bit; stream outputs and the debug spectrum of every call identical; ragged calls; a
    flush in mid-stream after which the stream goes on.  verbose = 1: the host engine (its submit is synchronous: one call late)."""
    nb = sum(sizes)
    H = N - N // R
    x = _bursty_stream(N, R, nb, 8, [(-0.2, 3, 9), (0.31, 6, 15), (-0.2, 17, 22), (0.33, 20, nb - 2)])
    args = (8, 1, N, R, [[0.1, 0.05]], [[-0.2, 0.04]], 6.0, 1.0, 0.0, 'normalized', 1,
            True, False, "", False, [[0.25, 0.4]], 10.0, 0.005, 1, 0.2, verbose, 0, 3, 3, True)
    cuts = np.cumsum([0] + sizes)
    serial = G.FrequencyDomainChannelizer(*args, max_blocks=max(sizes))
    piped = G.FrequencyDomainChannelizer(*args, max_blocks=max(sizes), pipelined=True)
    assert serial.pipeline.sinks_latency(serial.sinks) == 0
    lat = piped.pipeline.sinks_latency(piped.sinks)
    assert lat == (1 if verbose else 2) and piped.sinks.engine() == (0 if verbose else 1)
    ref_msgs, got_msgs = [], []
    halves = [range(0, len(sizes) // 2), range(len(sizes) // 2, len(sizes))]
    for part in halves:                                       # a flush in mid-stream, then the stream goes on through the same handles
        per_call = []
        for k, i in enumerate(part):
            xi = x[cuts[i] * H:cuts[i + 1] * H]
            rp = serial.work(xi)
            gp = piped.work(xi)
            for a, b in zip(gp, rp):                          # debug spectrum + the throughput channel: this call's items, now
                assert np.array_equal(a, b)
            ref_msgs += serial.messages
            per_call.append(len(serial.messages))
            n_before = len(got_msgs)
            got_msgs += piped.messages
            # the pipelined block hands out exactly the PDUs of the call `lat` calls ago
            assert len(got_msgs) - n_before == (per_call[k - lat] if k >= lat else 0)
        assert serial.flush() == []
        tail = piped.flush()
        assert len(tail) == sum(per_call[max(0, len(per_call) - lat):])
        got_msgs += tail
        assert piped.flush() == []
        _same_messages(got_msgs, ref_msgs)
    assert len(ref_msgs) >= 4
    capfd.readouterr()


@pytest.mark.parametrize("N,min_block_launch,lookahead", [(65536, 1, False), (65536, 1, True), (32768, 1, False), (16384, 1, True),
                                                          (65536, None, False), (4096, None, False)])
def test_power_cells_from_the_forward_kernels_group_sums(oracle, N, min_block_launch, lookahead):
    """Round 6: the forward kernel leaves the power of every 16-bin group of the spectrum beside the spectrum (fdc_pipeline_process_device_power) and
    the bank sums its cells from them plus the bins of the groups a cell cuts (fdc_sinks_prepare_from_groups) instead of reading the spectrum back.
    The block kernel's epilogue at N = 65536 / 32768 / 16384, the pass over the spectrum where another transform ran (short launch groups at the
    default threshold, N = 4096); cells that start and end anywhere inside a group (PowerActivationChannels) and cells of dec bins on the
    detector's grid; a bank fed through the spectrum pass gives the same PDUs (metadata equal, payloads bit for bit: the extractions do not
    depend on the cells), and both match the oracle."""
    R, sizes = 2, [9, 16, 3, 16]
    nb = sum(sizes)
    H = N - N // R
    carriers = [(-0.2, 3, 11), (0.31, 6, 17), (-0.2, 20, 30), (0.33, 25, nb - 3), (0.12, 2, 40)]
    x = _bursty_stream(N, R, nb, 8, carriers)
    pac = [((-0.2 + 0.5) % 1.0, 0.04, 0), ((0.12 + 0.5) % 1.0, 37.0 / N + 0.011, 1)]
    segs = [(0.75, 0.9)]
    kw = dict(pac=pac, pac_thresh=6.0, pac_maxblocks=3, segments=segs, det_thresh=10.0, det_maxblocks=3, minchandist=0.005, det_delay=1,
              puffer=0.2, max_blocks=16)
    hip = _HipCopy()
    hip.h.hipMalloc.argtypes = [hip.C.POINTER(hip.C.c_void_p), hip.C.c_size_t]
    ring = hip.C.c_void_p()
    assert hip.h.hipMalloc(hip.C.byref(ring), (N // R + nb * H) * 8) == 0
    pipe = G.Pipeline(N, R, [], windowtype=1, max_blocks=16, keep_spectrum=True, min_block_launch=min_block_launch)
    xr = np.concatenate([np.zeros(N // R, np.complex64), x])
    hip.upload(ring, xr, pipe.stream())
    cuts = np.cumsum([0] + sizes)
    results = []
    for from_groups in (True, False):
        bank = G.Sinks(N, R, lookahead=lookahead, **kw)
        assert bank.group_power_ptr() and (bool(bank.group_power_ahead_ptr()) == lookahead)
        got = []
        q = bank.fill_stream() if lookahead else bank.stream()

        def fill(i, ahead):
            pipe.process_device(int(ring.value) + 8 * int(cuts[i]) * H, int(cuts[i]), sizes[i], None,
                                d_spectrum=bank.spectrum_ahead_ptr() if ahead else bank.spectrum_ptr(), stream=q,
                                d_group_power=(bank.group_power_ahead_ptr() if ahead else bank.group_power_ptr()) if from_groups else None)
            if from_groups or lookahead:
                bank.prepare(sizes[i], ahead=ahead, from_groups=from_groups)
        if lookahead:
            fill(0, False)
        for i in range(len(sizes)):
            if lookahead:
                if i + 1 < len(sizes):
                    fill(i + 1, True)
            else:
                fill(i, False)
            got += bank.submit_device(sizes[i])
        got += bank.flush()
        results.append(got)
        bank.close()
    assert len(results[0]) == len(results[1]) >= 6
    for (gm, gd), (rm, rd) in zip(*results):
        assert {k: gm[k] for k in gm if k != "id"} == {k: rm[k] for k in rm if k != "id"}
        assert np.array_equal(gd, rd)
    _, spec = oracle.channelizer(N, R, 1, [], x, want_spectrum=True, nthreads=4)
    spec = spec.reshape(nb, N)
    for i, (cf, bw, ident) in enumerate(pac):
        compare([g for g in results[0] if g[0]["kind"] == 0 and g[0]["source"] == ident], oracle.PowerActivationChannel(N, cf, bw, R, 6.0, 3, 0, ident).work(spec), vec=False)
    compare([g for g in results[0] if g[0]["kind"] == 1], oracle.ActivityDetectionVcm(N, [list(segs[0])], 10.0, R, 3, 0.005, 1, 0.2).work(spec))
    hip.h.hipFree.argtypes = [hip.C.c_void_p]
    hip.h.hipFree(ring)


def test_a_batch_prepared_ahead_is_committed():
    """Round 6: on a look-ahead bank the submit in front of a batch that was prepared AHEAD enqueues that batch's decision chain at once.  From then
    on the batch is committed: feeding the bank from the host or preparing the current buffer again is refused (handle stays usable), the submit that
    follows must be for exactly that batch — another block count kills the handle instead of advancing the state machines twice."""
    N, R = 4096, 2
    spec = burst_spectrum(N, 24, [(1200, 1300, 3, 9, 1.0), (1200, 1300, 14, 20, 1.0)], 4)
    kw = dict(pac=[(1250 / N, 100 / N, 0)], pac_thresh=6.0, pac_maxblocks=3, max_blocks=8)
    ref = G.Sinks(N, R, **kw)
    want = []
    for a in (0, 8, 16):
        want += ref.work(spec[a:a + 8].reshape(-1))
    hip = _HipCopy()
    bank = G.Sinks(N, R, lookahead=True, **kw)
    fs = bank.fill_stream()
    b = [np.ascontiguousarray(spec[a:a + 8].reshape(-1)) for a in (0, 8, 16)]
    hip.upload(bank.spectrum_ptr(), b[0], fs)
    bank.prepare(8, ahead=False)
    hip.upload(bank.spectrum_ahead_ptr(), b[1], fs)
    bank.prepare(8, ahead=True)
    got = bank.submit_device(8)                            # batch 0; batch 1's chain goes out inside this call
    with pytest.raises(G.FdcError):
        bank.work(b[2])                                    # the committed batch first
    with pytest.raises(G.FdcError):
        bank.prepare(8, ahead=False)
    hip.upload(bank.spectrum_ahead_ptr(), b[2], fs)        # the ahead buffer is free for batch 2 meanwhile
    bank.prepare(8, ahead=True)
    got += bank.submit_device(8)                           # batch 1, as promised
    got += bank.submit_device(8)                           # batch 2
    got += bank.flush()
    assert len(got) == len(want) >= 2
    for (gm, gd), (rm, rd) in zip(got, want):
        assert {k: gm[k] for k in gm if k != "id"} == {k: rm[k] for k in rm if k != "id"} and np.array_equal(gd, rd)
    # the other block count: dead handle
    bank2 = G.Sinks(N, R, lookahead=True, **kw)
    hip.upload(bank2.spectrum_ptr(), b[0], bank2.fill_stream())
    bank2.prepare(8, ahead=False)
    hip.upload(bank2.spectrum_ahead_ptr(), b[1][:5 * N], bank2.fill_stream())
    bank2.prepare(5, ahead=True)
    bank2.submit_device(8)
    with pytest.raises(G.FdcError):
        bank2.submit_device(8)                             # 5 were prepared and are on their way
    with pytest.raises(G.FdcError):
        bank2.submit_device(5)                             # dead from the first wrong call on


def test_randomised_hier_block_pipelined_equals_serial(monkeypatch, capsys):
    """tools/fuzz_hier.py, twelve seeded cases inside the suite: random block lengths, overlaps, throughput / activity-controlled channels, detection
    segments, maxblocks, delays, call cuts and flushes in mid-stream — the pipelined hier block (group sums, early decision chains, PDUs two calls late)
    against the serial form bit for bit, and both against a bank fed the debug spectrum (cells by the pass over the spectrum)."""
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fuzz_hier", os.path.join(root, "tools", "fuzz_hier.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(sys, "argv", ["fuzz_hier.py", "12", "2026"])
    mod.main()
    assert "12 cases" in capsys.readouterr().out
