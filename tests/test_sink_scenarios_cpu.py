"""CPU tests (-m "not gpu"): the ORACLE's sink restatements against the hand-computed scenarios of tests/sink_scenarios.py
(expected PDUs derived on paper from the reference's text, each with the lines it follows)."""
import pytest

import sink_scenarios as S


def _pairs(pdus):
    return [(d, d["samples"]) for d in pdus]


@pytest.mark.parametrize("sc", S.VCM, ids=[s["name"] for s in S.VCM])
def test_oracle_vcm(oracle, sc):
    blk = oracle.ActivityDetectionVcm(S.N, [sc.get("segment", S.SEG)], 10.0, S.R, sc["maxblocks"], 0.0625, sc["delay"], sc["puffer"])
    S.check(sc["name"], _pairs(blk.work(sc["spec"])), sc["expect"])


@pytest.mark.parametrize("sc", S.PAC, ids=[s["name"] for s in S.PAC])
def test_oracle_pac(oracle, sc):
    blk = oracle.PowerActivationChannel(S.N, 0.5, 16.0 / S.N, S.R, 6.0, sc["maxblocks"], 0, 9)
    assert (blk.extract_start, blk.extract_stop, blk.measure_start, blk.measure_stop, blk.output_len) == (120, 136, 120, 136, 8)
    S.check(sc["name"], _pairs(blk.work(sc["spec"])), sc["expect"])


@pytest.mark.parametrize("sc", S.SD, ids=[s["name"] for s in S.SD])
def test_oracle_segment_detection(oracle, sc):
    ident, a, b = sc["sd"]
    blk = oracle.SegmentDetection(ident, S.N, S.R, a, b, 10.0, 0.0625, sc["puffer"], sc["maxblocks"], sc["delay"])
    assert blk.segments[0] == sc["geometry"]
    S.check(sc["name"], _pairs(blk.work(sc["spec"])), sc["expect"])


@pytest.mark.parametrize("sc", S.PAC_GEOM, ids=[s["name"] for s in S.PAC_GEOM])
def test_oracle_pac_geometry_and_payload(oracle, sc):
    cf, bw = sc["pac"]
    blk = oracle.PowerActivationChannel(S.N, cf, bw, S.R, 6.0, sc["maxblocks"], 0, 9)
    assert (blk.extract_start, blk.extract_stop, blk.measure_start, blk.measure_stop, blk.output_len) == sc["params"]
    S.check(sc["name"], _pairs(blk.work(sc["spec"])), sc["expect"])
