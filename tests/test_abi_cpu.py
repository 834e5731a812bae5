"""CPU tests (-m "not gpu") of the product's host side: the C-ABI library loads and exports every
symbol include/fdc_amd.h declares, host-side window design and parameter derivation match the
reference fixtures, and nothing silently computes on the CPU when no GPU is present."""
import json
import os
import re

import numpy as np
import pytest

import gr_fdc_amd as G
from gr_fdc_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "fdc_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(fdc_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    h = G.lib()
    for name in sorted(declared):
        assert hasattr(h, name), "libfdc_amd.so does not export " + name
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)


def test_no_oracle_in_product_sources():
    """The product path must not import, link or call the oracle."""
    pkg = os.path.join(ROOT, "gr-fdc_amd")
    for dp, _dn, fn in os.walk(pkg):
        for f in fn:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h", "Makefile")):
                txt = open(os.path.join(dp, f)).read()
                assert "fdc_oracle" not in txt and "fdco_" not in txt and "oracle." not in txt.replace("oracle/", ""), f


def test_window_table_bit_exact_vs_reference_fixture(golden_dir):
    """fdc_window_table (product, host-side design) == the reference's own cr_win tables."""
    z = np.load(os.path.join(golden_dir, "windows_ref.npz"))
    for i, (t, l, p, s, R) in enumerate(z["params"]):
        w = G.window_table(int(t), int(l), np.float32(p), np.float32(s), int(R))
        assert (w.view(np.uint32) == z["case%02d" % i].view(np.uint32)).all(), i


def _params_fixture(golden_dir):
    return json.load(open(os.path.join(golden_dir, "channel_params.json")))


def test_get_opt_channelparams_vs_reference_fixture(golden_dir):
    """Product parameter derivation == the reference's own get_opt_channelparams on every row of the fixture that
    tests/golden/make_params_from_reference.py produced by running python/FrequencyDomainChannelizer.py:322-345
    (>= 1500 points: BASELINE plans, a random sweep, .5 ties, clamp / wrap edges, passband < 0.7, both fs modes)."""
    fx = _params_fixture(golden_dir)
    assert len(fx["rows"]) >= 1000 and "make_params_from_reference.py" in fx["generated_by"]
    nties = 0
    for r in fx["rows"]:
        _m, get_freq, _sf, get_bw, _sb = G.freq_converters(r.get("freqmode", 0), r.get("fs", 1.0), r.get("centerfrequency", 0.0))
        fr, bw = get_freq(r["freq"]), get_bw(r["bw"])
        assert [fr, bw] == r["internal"], r                       # the mode lambdas (:70-91), bit for bit
        N, R = G.nextpow2(r["N"]), G.nextpow2(r["R"])             # :138-139
        if isinstance(r["out"], dict):
            with pytest.raises(ValueError):
                G.get_opt_channelparams(N, R, fr, bw)
            continue
        got = G.get_opt_channelparams(N, R, fr, bw)
        assert list(got) == r["out"], (r, got)                    # ints exact, doubles bit-identical
        if r.get("tie"):
            nties += 1
            assert list(got) != r["out_py3"]                      # Python-2 rounding is what the reference runs under
    assert nties >= 5


def test_nextpow2_vs_reference_fixture(golden_dir):
    fx = _params_fixture(golden_dir)
    for k, v in fx["nextpow2"]:
        assert G.nextpow2(k) == v, k
    assert fx["nextpow2_below_one_raises"]
    with pytest.raises(ValueError):
        G.nextpow2(0.5)


def test_frequency_modes_vs_reference_fixture(golden_dir):
    """normalized / basebandfs / centerfreqfs (integer and string spellings): channel and segment conversion and the
    inverse lambdas, against what the reference's __init__ (:70-91, :349-357) produced."""
    fx = _params_fixture(golden_dir)
    assert len(fx["modes"]) == 6
    for m in fx["modes"]:
        mode, get_freq, set_freq, get_bw, set_bw = G.freq_converters(m["freqmode"], m["fs"], m["centerfrequency"])
        assert mode == m["freqmode_int"]
        assert [[get_freq(a), get_bw(b)] for a, b in m["channels"]] == m["throughput_channels"]
        assert [[get_freq(a), get_freq(b)] for a, b in m["segments"]] == m["activity_detection_segments"]
        assert [set_freq(v) for v in (0.0, 0.25, 0.5, 0.75)] == m["set_freq"]
        assert [set_bw(v) for v in (0.01, 0.5)] == m["set_bw"]
    with pytest.raises(ValueError):
        G.freq_converters(3)
    with pytest.raises(ValueError):
        G.freq_converters("hz")


def test_channelparams_match_oracle_on_a_sweep(oracle):
    rng = np.random.default_rng(0)
    for _ in range(500):
        N = int(2 ** rng.integers(6, 19)); R = int(2 ** rng.integers(1, 4))
        fr = float(rng.uniform(0, 1)); bw = float(rng.uniform(2.0 / N, 0.3))
        assert G.get_opt_channelparams(N, R, fr, bw) == oracle.channel_params(N, R, fr, bw)


@pytest.mark.skipif(G.lib().fdc_device_count() > 0, reason="a GPU is present")
def test_fails_loudly_without_gpu():
    with pytest.raises(G.FdcError) as ei:
        G.Pipeline(4096, 2, [(0, 256, 0.88, 1.0)])
    assert "FDC_ERR_NO_DEVICE" in str(ei.value)
    with pytest.raises(G.FdcError):
        G.overlap_save(8, 16, 4)
    with pytest.raises(G.FdcError):
        G.fft_vcc(16, True, True, np.zeros(16, np.complex64))


def test_argument_validation_precedes_device_use():
    # predicates of the reference ctors (lib/phase_shifting_windowing_vcc_impl.cc:46-53) -> ValueError
    with pytest.raises(ValueError):
        G.phase_shifting_windowing_vcc(64, 2, 0, 0.9, 0.5, 1)
    with pytest.raises(ValueError):
        G.phase_shifting_windowing_vcc(64, 2, 0, 0.0, 0.5, 1)
    with pytest.raises(ValueError):
        G.Pipeline(4096, 2, [(4000, 256, 0.88, 1.0)])      # slice leaves the spectrum
    with pytest.raises(ValueError):
        G.Pipeline(1000, 2, [])                            # not a power of two


def test_header_is_plain_c_and_example_compiles(tmp_path):
    """include/fdc_amd.h is a C header (C99, -pedantic) and the plain-C example builds against it without hipcc."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    obj = str(tmp_path / "ex.o")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"),
                           "-c", os.path.join(root, "examples", "fdc_pipeline_example.c"), "-o", obj])


def test_round_half_away_is_c_round_on_the_edges():
    """get_opt_channelparams rounds like the Python-2 reference (half away from zero), which is C's round(): the product's
    helper must agree with libm on the values where `floor(abs(x) + 0.5)` does not."""
    import ctypes
    import ctypes.util
    from gr_fdc_amd.channelizer import _round_half_away
    m = ctypes.CDLL(ctypes.util.find_library("m"))
    m.round.restype = ctypes.c_double
    m.round.argtypes = [ctypes.c_double]
    for x in (0.49999999999999994, -0.49999999999999994, 0.5, -0.5, 1.5, 2.5, -2.5, 4503599627370497.0, -4503599627370497.0,
              4503599627370495.5, 1e300, 0.0, 123.49999999999999, 123.5):
        assert _round_half_away(x) == m.round(x), x


def test_exception_barrier_of_the_c_abi():
    """No C++ exception crosses the extern "C" boundary: the barrier every entry runs behind turns bad_alloc /
    system_error into FDC_ERR_NOMEM and anything else into FDC_ERR_HIP; and every int-returning entry that has a body of its
    own in the three ABI sources goes through that barrier (one-line accessors cannot throw)."""
    assert G.lib().fdc_selftest_exception_barrier() == 0, G.lib().fdc_last_error()
    csrc = os.path.join(ROOT, "gr-fdc_amd", "csrc")
    for fn in ("fdc_api.hip", "fdc_sinks.hip", "fdc_group.hip"):
        txt = open(os.path.join(csrc, fn)).read()
        ext = txt[txt.index('extern "C" {'):]
        for m in re.finditer(r"^int (fdc_\w+)\([^;{]*\)\n\{\n(.*?)^\}", ext, flags=re.S | re.M):
            name, body = m.group(1), m.group(2)
            if name == "fdc_selftest_exception_barrier" or body.count("\n") <= 8 and "std::" not in body and "new " not in body:
                continue
            assert "FDC_ENTRY(" in body, "%s: %s has no exception barrier" % (fn, name)


def test_group_entry_points_validate_without_a_device():
    """The multi-device handle refuses bad member lists before any device is touched."""
    import ctypes as C
    cfg = _lib.fdc_pipeline_cfg(0, 4096, 2, 1, 0, None, 8, 0, 0, 0, 0, 0)
    h = C.c_void_p()
    assert G.lib().fdc_pipeline_group_create(C.byref(cfg), None, 2, 0, C.byref(h)) == -1
    devs = (C.c_int32 * 2)(0, 0)
    assert G.lib().fdc_pipeline_group_create(C.byref(cfg), devs, 0, 0, C.byref(h)) == -1
    assert G.lib().fdc_pipeline_group_create(C.byref(cfg), devs, 65, 0, C.byref(h)) == -1
    assert G.lib().fdc_pipeline_group_size(None) == -1 and not G.lib().fdc_pipeline_group_member(None, 0)
    G.lib().fdc_pipeline_group_destroy(None)


def test_a_call_may_not_produce_4_gib_of_output():
    """Several kernels address one call's output with 32-bit byte offsets: max_blocks x (kept samples per block) x 8 bytes below 4 GiB is checked
    before any device is touched (the headline plan: 32768 samples per block -> at most 16383 blocks per call)."""
    import ctypes as C
    chans = (_lib.fdc_channel * 256)(*[_lib.fdc_channel(256 * c, 256, 0.88, 1.0) for c in range(256)])
    h = C.c_void_p()
    cfg = _lib.fdc_pipeline_cfg(0, 65536, 2, 1, 256, chans, 16384, 0, 0, 0, 0, 0)
    assert G.lib().fdc_pipeline_create(C.byref(cfg), C.byref(h)) == -1
    assert b"4 GiB" in G.lib().fdc_last_error() and b"16383" in G.lib().fdc_last_error()


def test_group_workers_are_pinned_to_their_devices_numa_node():
    """the worker thread of a group member runs on the CPUs of the NUMA node its device hangs off.  The placement code
    needs no device: fdc_selftest_worker_placement starts a worker exactly as a group does for a member on `node` and reports the mask it
    runs under.  Every online node: pinned inside the node's CPUs (or left alone when none of them is in this process's cpuset); an unknown
    node (-1, 4095): left alone under the process's own mask.  Skips where sysfs lists no nodes."""
    import ctypes as C
    import gr_fdc_amd as G
    lib = G.lib()
    mine = os.sched_getaffinity(0)
    for bogus in (-1, 4095):
        a, b = C.c_int32(-1), C.c_int32(-1)
        assert lib.fdc_selftest_worker_placement(bogus, C.byref(a), C.byref(b)) == 0 and a.value == 0 and b.value == len(mine)
    assert lib.fdc_device_numa_node(-1) == -1 and lib.fdc_device_numa_node(10 ** 6) == -1
    try:
        online = open("/sys/devices/system/node/online").read().strip()
    except OSError:
        pytest.skip("no NUMA nodes in sysfs")
    nodes = []
    for part in online.split(","):
        lo, _, hi = part.partition("-")
        nodes += list(range(int(lo), int(hi or lo) + 1))
    checked = 0
    for node in nodes:
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            if part:
                lo, _, hi = part.partition("-")
                cpus |= set(range(int(lo), int(hi or lo) + 1))
        a, b = C.c_int32(-1), C.c_int32(-1)
        rc = lib.fdc_selftest_worker_placement(node, C.byref(a), C.byref(b))
        usable = cpus & mine
        assert rc == (1 if usable else 0), (node, rc, lib.fdc_last_error())
        assert a.value == len(usable) and b.value == (len(usable) if usable else len(mine))
        checked += 1
    assert checked >= 1
    assert os.sched_getaffinity(0) == mine            # the calling thread was never touched


def test_round6_entry_points_validate_without_a_device():
    """The entries added in round 6 refuse null handles before anything touches a device (and say so in fdc_last_error)."""
    lib = G.lib()
    assert lib.fdc_pipeline_flush_sinks(None, None) == -1 and b"null" in lib.fdc_last_error()
    assert lib.fdc_pipeline_sinks_latency(None, None) == -1
    assert lib.fdc_pipeline_process_device_power(None, None, 0, 1, None, None, None, None) == -1
    assert lib.fdc_sinks_prepare_from_groups(None, 1, 0) == -1
    assert not lib.fdc_sinks_group_power(None) and not lib.fdc_sinks_group_power_ahead(None)
    assert lib.fdc_pipeline_work_sinks(None, None, 1, None, None, None) == -1
