"""GPU tests (-m gpu): the HIP sink engine, through the C-ABI, against the hand-computed scenarios of
tests/sink_scenarios.py (expected PDUs derived on paper from the reference's text; the oracle is not involved here)."""
import pytest

import gr_fdc_amd as G
import sink_scenarios as S

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sc", S.VCM, ids=[s["name"] for s in S.VCM])
def test_hip_vcm(sc):
    blk = G.activity_detection_channelizer_vcm(S.N, [sc.get("segment", S.SEG)], 10.0, S.R, sc["maxblocks"], False, False, "", False,
                                               0.0625, sc["delay"], sc["puffer"], 0, max_blocks=5)
    S.check(sc["name"], blk.bank.work(sc["spec"]), sc["expect"])          # 12 items in batches of 5: state crosses the calls
    whole = G.activity_detection_channelizer_vcm(S.N, [sc.get("segment", S.SEG)], 10.0, S.R, sc["maxblocks"], False, False, "", False,
                                                 0.0625, sc["delay"], sc["puffer"], 0, max_blocks=16)
    S.check(sc["name"] + " (one batch)", whole.bank.work(sc["spec"]), sc["expect"])


@pytest.mark.parametrize("sc", S.PAC, ids=[s["name"] for s in S.PAC])
def test_hip_pac(sc):
    for mb in (3, 16):
        blk = G.PowerActivationChannel(S.N, 0.5, 16.0 / S.N, S.R, 6.0, sc["maxblocks"], 0, False, False, "", 0, 9, max_blocks=mb)
        p = blk.params
        assert (p["extract_start"], p["extract_stop"], p["measure_start"], p["measure_stop"], p["output_len"]) == (120, 136, 120, 136, 8)
        got = blk.bank.work(sc["spec"])
        S.check(sc["name"], got, sc["expect"])
        assert all(m["source"] == 9 for m, _s in got)
        # the ID string carries the running number of the activation (PowerActivationChannel_impl.cc:308-312)
        assert all(m["id"].endswith(".PowActChan.9.%d" % m["chan_id"]) for m, _s in got)


@pytest.mark.parametrize("engine", ["device", "host"])
@pytest.mark.parametrize("sc", S.PAC_GEOM, ids=[s["name"] for s in S.PAC_GEOM])
def test_hip_pac_geometry_and_payload(sc, engine):
    cf, bw = sc["pac"]
    for mb in (3, 16):
        bank = G.Sinks(S.N, S.R, pac=[(cf, bw, 9)], pac_thresh=6.0, pac_maxblocks=sc["maxblocks"], max_blocks=mb,
                       host_decisions=engine == "host")
        p = bank.pac_params(0)
        assert (p["extract_start"], p["extract_stop"], p["measure_start"], p["measure_stop"], p["output_len"]) == sc["params"]
        S.check(sc["name"] + " / " + engine, bank.work(sc["spec"]), sc["expect"])


@pytest.mark.parametrize("sc", S.SD, ids=[s["name"] for s in S.SD])
def test_hip_segment_detection(sc):
    ident, a, b = sc["sd"]
    blk = G.SegmentDetection(ident, S.N, S.R, a, b, 10.0, 0.0625, sc["puffer"], sc["maxblocks"], sc["delay"], False, False, "", False, 0,
                             max_blocks=7)
    assert blk.segment == sc["geometry"]
    got = blk.bank.work(sc["spec"])
    S.check(sc["name"], got, sc["expect"])
    assert all(m["id"].endswith(".DETECTED.%d.%d" % (ident, m["chan_id"])) for m, _s in got)
