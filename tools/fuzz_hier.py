#!/usr/bin/env python3
"""GPU box helper (round 6): randomized differential test of the PIPELINED hier block (fdc_pipeline_work_sinks on a look-ahead bank: copy and forward
transform of call n beside the sinks of call n - 1, power cells from the forward kernel's group sums, the prepared batch's decision chain enqueued early,
PDUs two calls late, fdc_pipeline_flush_sinks) against the SERIAL form fed the same stream in the same calls, and of both against a bank fed with the
debug spectrum of the serial form through fdc_sinks_work (power cells by the pass over the spectrum).  Every case draws a block length, an overlap,
throughput channels, activity-controlled channels, detection segments, thresholds, maxblocks, delays, a stream of carriers keyed on and off at random
and a pattern of calls with flushes in mid-stream.  Everything must agree exactly between the two forms: order, metadata, payload bits, stream outputs;
the third bank's PDUs must carry the same metadata (its cells are summed in another order: payloads are compared to 1e-6).
Usage: python tools/fuzz_hier.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gr_fdc_amd as G  # noqa: E402
from gr_fdc_amd.sinks import _pac_pdu, _det_pdu  # noqa: E402


def stream(rng, N, R, nb, carriers):
    H = N - N // R
    n = np.arange(nb * H)
    x = 0.01 * (rng.standard_normal(nb * H) + 1j * rng.standard_normal(nb * H))
    for fc, bw in carriers:
        env = np.zeros(nb * H)
        m, on = 0, bool(rng.integers(0, 2))
        while m < nb:
            ln = int(rng.integers(2, 12))
            if on:
                env[m * H:(m + ln) * H] = 1.0
            m += ln
            on = not on
        sps = max(2, int(round(1.0 / max(bw, 1e-4))))
        sym = (rng.integers(0, 2, nb * H // sps + 2) * 2 - 1) + 1j * (rng.integers(0, 2, nb * H // sps + 2) * 2 - 1)
        x += env * np.repeat(sym, sps)[:nb * H] * np.exp(2j * np.pi * fc * n)
    return x.astype(np.complex64)


def same(a, b, bits=True):
    if len(a) != len(b):
        return "%d PDUs against %d" % (len(a), len(b))
    for i, ((da, sa), (db, sb)) in enumerate(zip(a, b)):
        ka = {k: v for k, v in da.items() if k != "ID"}
        kb = {k: v for k, v in db.items() if k != "ID"}
        if ka != kb or da["ID"][20:] != db["ID"][20:]:
            return "PDU %d: %s against %s" % (i, da, db)
        if bits:
            if not np.array_equal(sa, sb):
                return "PDU %d: payload bits differ" % i
        elif sa.size != sb.size or (sa.size and np.abs(sa - sb).max() > 1e-6 * max(1e-30, np.abs(sb).max())):
            return "PDU %d: payload differs" % i
    return None


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    npdu = 0
    for case in range(cases):
        N = int(rng.choice([4096, 16384, 65536, 32768, 8192]))
        R = int(rng.choice([2, 4]))
        nb = int(rng.integers(20, 70))
        carriers = [(float(rng.uniform(-0.45, 0.45)), float(rng.uniform(0.004, 0.03))) for _ in range(int(rng.integers(2, 7)))]
        x = stream(rng, N, R, nb, carriers)
        thr = [[float(rng.uniform(-0.4, 0.4)), float(rng.uniform(0.01, 0.05))] for _ in range(int(rng.integers(0, 3)))]
        pick = [c for c in carriers if -0.48 < c[0] - c[1] and c[0] + c[1] < 0.48]
        acc = [[c[0], min(0.08, c[1] * float(rng.uniform(1.0, 2.5)))] for c in pick[:int(rng.integers(0, 3))]]
        segs = []
        if rng.integers(0, 3) or not acc:
            a = float(rng.uniform(-0.45, 0.2))
            segs.append([a, a + float(rng.uniform(0.1, 0.25))])
        mb = int(rng.choice([-1, 0, 1, 3, 7]))
        args = (8, 1, N, R, thr, acc, 6.0, 1.0, 0.0, 'normalized', 1, True, False, "", False, segs, 10.0, float(rng.choice([0.005, 0.01])),
                int(rng.integers(0, 3)), 0.2, 0, int(rng.integers(0, 3)), mb, mb, True)
        maxb = int(rng.choice([8, 16, 24]))
        serial = G.FrequencyDomainChannelizer(*args, max_blocks=maxb)
        piped = G.FrequencyDomainChannelizer(*args, max_blocks=maxb, pipelined=True)
        H = N - N // R
        ref, got, specs = [], [], []
        m = 0
        while m < nb:
            k = int(min(nb - m, rng.integers(1, maxb + 1)))
            rp = serial.work(x[m * H:(m + k) * H])
            gp = piped.work(x[m * H:(m + k) * H])
            for a, b in zip(gp, rp):
                assert np.array_equal(a, b), "case %d: stream outputs differ" % case
            specs.append(rp[0])
            ref += serial.messages
            got += piped.messages
            if rng.integers(0, 6) == 0:
                got += piped.flush()                       # a flush in mid-stream
                assert len(got) == len(ref), "case %d: after a flush the pipelined form has handed out %d of %d PDUs" % (case, len(got), len(ref))
            m += k
        got += piped.flush()
        why = same(got, ref)
        assert why is None, "case %d (N %d R %d): pipelined against serial: %s" % (case, N, R, why)
        # the sink blocks alone on the serial form's debug spectrum: cells by the pass over the spectrum
        third = []
        if serial.sinks is not None:
            from gr_fdc_amd.sinks import Sinks
            kw = dict(pac=[(cf, bw, i) for i, (cf, bw) in enumerate(serial.activity_controlled_channels)], pac_thresh=6.0, pac_maxblocks=mb,
                      pac_delay=args[21], segments=[tuple(s) for s in serial.activity_detection_segments], det_thresh=10.0, det_maxblocks=mb,
                      minchandist=serial.get_bw(args[17]) if segs else 0.005, det_delay=args[18], puffer=0.2, max_blocks=maxb, det_variant=1)
            bank = Sinks(N, R, **kw)
            for sp in specs:
                raw = bank.work(sp)
                third += [_pac_pdu(mm, d) for (mm, d) in raw if mm["kind"] == 0] + [_det_pdu(mm, d) for (mm, d) in raw if mm["kind"] == 1]
            why = same(third, ref, bits=False)
            assert why is None, "case %d (N %d R %d): bank on the debug spectrum against the hier block: %s" % (case, N, R, why)
        npdu += len(ref)
        print("case %3d  N %6d R %d  %2d blocks  %d thr %d acc %d seg  maxblocks %2d  %4d PDUs  ok" % (case, N, R, nb, len(thr), len(acc), len(segs), mb, len(ref)))
    print("fuzz_hier: %d cases, %d PDUs: pipelined == serial bit for bit, both == the bank on the debug spectrum" % (cases, npdu))


if __name__ == "__main__":
    main()
