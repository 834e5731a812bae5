"""Hand-computed scenarios for the stateful sink blocks, derived directly from the reference's text (NOT from either
restatement): every expected PDU below was worked out on paper from the cited lines of lib/PowerActivationChannel_impl.cc,
lib/activity_detection_channelizer_vcm_impl.cc and lib/SegmentDetection_impl.cc.  The same table is run against the oracle
(tests/test_sink_scenarios_cpu.py) and against the HIP product through the C-ABI (tests/test_sink_scenarios_gpu.py), so a
misreading shared by the two sibling restatements has something to fail against.

Inputs are spectrum items with EXACT per-cell powers (amplitudes 1, 4, 8, 10, 16, 32: squares and their sums of 8 or 16 equal
terms are exact in float32 in any summation order), so no decision depends on rounding.

Detection geometry used throughout (…vcm_impl.cc:230-279): N = 256, relinvovl = 2, minchandist = 0.0625 -> decimation
int(256 * 0.0625 / 2) = 8 (:234-240); segment [0.125, 0.875] -> mid = round(0.5 * 256) = 128, width = round(0.75 * 256) = 192
(a multiple of 8), start = 128 - 96 = 32, stop = 224, 24 power cells; cell i covers bins [32 + 8 i, 40 + 8 i).
Threshold 10 dB -> 10.0 (:119).  Edge positions (:708-709): a rising edge between cells i-1 and i sits at (i - 1) * 8 + start,
a falling edge at i * 8 + start.  Block counter: vcm starts at 1 (:188), so during input block m it reads m + 1;
SegmentDetection starts at 0 (SegmentDetection_impl.cc:118); PowerActivationChannel at 1 (PowerActivationChannel_impl.cc:96).
"""
import numpy as np

N, R = 256, 2
SEG = [0.125, 0.875]
START, DEC = 32, 8


def cell_spectrum(nb, bursts, floor=1.0, start=START):
    """nb items; power `floor` in every bin, then rectangular bursts (first cell, last cell + 1, first block, last block,
    power per bin).  Cells are 8 bins wide from `start`."""
    s = np.full((nb, N), np.sqrt(floor), dtype=np.float64)
    for c0, c1, b0, b1, p in bursts:
        s[b0:b1 + 1, start + DEC * c0:start + DEC * c1] = np.sqrt(p)
    return s.astype(np.complex64)


def pdu(chan, fin, part, bs, be, vs, ve, ns):
    """part = None: the dict carries no "part" key (…vcm_impl.cc:419-420)."""
    return dict(chan_id=chan, finalized=fin, part=part, blockstart=bs, blockend=be, vectorstart=vs, vectorend=ve, nsamples=ns)


# A burst on cells 5..9 in blocks 3..6: rising edge at i = 5 -> (5-1)*8+32 = 64, falling edge at i = 10 -> 10*8+32 = 112.
# activate (:785-841): width 48, mid 64 + 24 = 88, extract width nextpow2(48) = 64, extract [56, 120), 64 - 64/2 = 32 samples
# per block.  Block 3: inactive < 0 -> previous + current block (count 2, :399-403); blocks 4, 5, 6: count 3, 4, 5.
# Block 7: no candidate -> inactive 1.  delay 0: 1 > 0 -> emitted in block 7 (:309): blockend 7+1, blockstart 8-5.
# delay 1: block 7 is still processed (count 6), block 8: inactive 2 > 1 -> emitted: blockend 9, blockstart 9-6.
# delay 2: blocks 7 and 8 processed (count 7), block 9: inactive 3 > 2: blockend 10, blockstart 10-7.
_B = [(5, 10, 3, 6, 256.0)]
VCM = [
    dict(name="vcm inactive > delay, delay 0", delay=0, maxblocks=-1, puffer=0.0, spec=cell_spectrum(12, _B),
         expect=[pdu(0, True, None, 3, 8, 56, 120, 5 * 32)]),
    dict(name="vcm inactive == delay keeps the channel one more block, delay 1", delay=1, maxblocks=-1, puffer=0.0,
         spec=cell_spectrum(12, _B), expect=[pdu(0, True, None, 3, 9, 56, 120, 6 * 32)]),
    dict(name="vcm delay 2", delay=2, maxblocks=-1, puffer=0.0, spec=cell_spectrum(12, _B),
         expect=[pdu(0, True, None, 3, 10, 56, 120, 7 * 32)]),
    # Two rising edges of EQUAL ratio 16 (cells 2-3 at 16, cells 4-7 at 256): positions (2-1)*8+32 = 40 and (4-1)*8+32 = 56, one
    # falling edge at 8*8+32 = 96.  std::sort (:713) leaves the order of equal keys unspecified; libstdc++ sorts ranges this
    # short by insertion, which keeps them in order: (40, 96) is taken first, (56, 96) overlaps it (:727-734) and is dropped.
    # width 56 -> extract width 64, mid 68, extract [36, 100).
    dict(name="vcm tie of two equal rising edges: the lower one first", delay=1, maxblocks=-1, puffer=0.0,
         spec=cell_spectrum(12, [(2, 4, 3, 6, 16.0), (4, 8, 3, 6, 256.0)]),
         expect=[pdu(0, True, None, 3, 9, 36, 100, 6 * 32)]),
    # A falling edge exactly AT a rising position: cells 4-5 at 1024, cell 6 at 1, cells 7-9 at 256.  Rising edges (1024 -> 56),
    # (256 -> (7-1)*8+32 = 80); falling edges at 6*8+32 = 80 and 10*8+32 = 112.  get_next_int (:678-692) wants a falling edge
    # strictly above the rising one: (56, 80) and, for the edge at 80, not 80 but 112 -> (80, 112); 80 < 80 is false, so the
    # two do not overlap (:727).  Strongest first: channel 0 = (56, 80): width 24 -> 32, mid 68, [52, 84); channel 1 = (80, 112):
    # width 32 -> 32, mid 96, [80, 112).  16 samples per block each.
    dict(name="vcm falling edge at a rising position", delay=1, maxblocks=-1, puffer=0.0,
         spec=cell_spectrum(12, [(4, 6, 3, 6, 1024.0), (7, 10, 3, 6, 256.0)]),
         expect=[pdu(0, True, None, 3, 9, 52, 84, 6 * 16), pdu(1, True, None, 3, 9, 80, 112, 6 * 16)]),
    # Overlapping candidates: cells 2-3 at 16, cells 4-5 at 1024: rising edges (16 -> 40) and (64 -> 56), one falling edge at
    # 6*8+32 = 80.  The stronger edge is served first (:713): (56, 80); then (40, 80): 40 < 80 and 80 >= 56 -> overlap, dropped.
    dict(name="vcm candidate overlapping an accepted one is dropped", delay=1, maxblocks=-1, puffer=0.0,
         spec=cell_spectrum(12, [(2, 4, 3, 6, 16.0), (4, 6, 3, 6, 1024.0)]),
         expect=[pdu(0, True, None, 3, 9, 52, 84, 6 * 16)]),
    # maxblocks 0: everything buffered goes out in every block (:317-318, :454-470), "part" counts up, blockstart = counter -
    # count keeps pointing at block 2 while the channel lives; the final PDU (block 8) is empty, carries part 5 (:419-420) and
    # reads 9 - 6.
    dict(name="vcm maxblocks 0", delay=1, maxblocks=0, puffer=0.0, spec=cell_spectrum(12, _B),
         expect=[pdu(0, False, 0, 2, 4, 56, 120, 64), pdu(0, False, 1, 2, 5, 56, 120, 32), pdu(0, False, 2, 2, 6, 56, 120, 32),
                 pdu(0, False, 3, 2, 7, 56, 120, 32), pdu(0, False, 4, 2, 8, 56, 120, 32), pdu(0, True, 5, 3, 9, 56, 120, 0)]),
    # maxblocks 1: one block per PDU, always one block behind (two are buffered at activation); the final PDU has the last one
    dict(name="vcm maxblocks 1", delay=1, maxblocks=1, puffer=0.0, spec=cell_spectrum(12, _B),
         expect=[pdu(0, False, 0, 2, 4, 56, 120, 32), pdu(0, False, 1, 2, 5, 56, 120, 32), pdu(0, False, 2, 2, 6, 56, 120, 32),
                 pdu(0, False, 3, 2, 7, 56, 120, 32), pdu(0, False, 4, 2, 8, 56, 120, 32), pdu(0, True, 5, 3, 9, 56, 120, 32)]),
    # Zero power everywhere but the burst: P[i-1] == 0 -> P[i] / FLT_MIN (:703-706): 256 / FLT_MIN is a rising edge at 64,
    # 0 / FLT_MIN = 0 a falling edge at every other cell, one of them AT 64 (i = 4 -> 4*8+32), which get_next_int skips.
    dict(name="vcm zero-power guard", delay=1, maxblocks=-1, puffer=0.0, spec=cell_spectrum(12, _B, floor=0.0),
         expect=[pdu(0, True, None, 3, 9, 56, 120, 6 * 32)]),
    # Extraction clamped at the band edge (:812-819): segment [0, 0.75] -> mid 96, width 192, start 0; burst on cells 1-2 ->
    # edges at 0 and 3*8 = 24; width 24, puffer 0.5 -> nextpow2(ceil(24 * 2)) = 64, mid 12, 12 - 32 < 0 -> [0, 64).
    dict(name="vcm extraction clamped at bin 0", delay=1, maxblocks=-1, puffer=0.5, segment=[0.0, 0.75],
         spec=cell_spectrum(12, [(1, 3, 3, 6, 256.0)], start=0), expect=[pdu(0, True, None, 3, 9, 0, 64, 6 * 32)]),
    # Carrier wider than a block after the flank puffer (:793-803): cells 1-22 -> edges 32 and 23*8+32 = 216, width 184,
    # puffer 0.25 -> ceil(276) -> 512 > 256: logged and skipped, nothing is ever emitted.
    dict(name="vcm carrier wider than the block is skipped", delay=1, maxblocks=-1, puffer=0.25,
         spec=cell_spectrum(12, [(1, 23, 3, 6, 256.0)]), expect=[]),
    # maxblocks 3 (:317-318, :454-470): block 3 buffers two blocks (count 2); block 4: count 3, three buffered -> the first three
    # go out, part 0, counter 5: blockstart 5 - 3; blocks 5, 6: one, two buffered; block 7 (inactive 1 <= delay) is still
    # processed: count 6, three buffered -> part 1, counter 8: 8 - 6; block 8: inactive 2 > 1 -> final PDU with NOTHING in it,
    # carrying part 2 (:419-420), counter 9: 9 - 6; the size check behind it (:317) sees 0 >= 3: nothing.
    dict(name="vcm maxblocks 3: the final PDU is empty", delay=1, maxblocks=3, puffer=0.0, spec=cell_spectrum(12, _B),
         expect=[pdu(0, False, 0, 2, 5, 56, 120, 96), pdu(0, False, 1, 2, 8, 56, 120, 96), pdu(0, True, 2, 3, 9, 56, 120, 0)]),
    # Two channels, maxblocks 3, delay 1 — the ORDER inside one block.  Channel 0: cells 5-9, blocks 2..11 -> (64, 112), extract
    # [56, 120), activated in block 2; its partial PDUs come at counts 3, 6, 9 = blocks 3, 6, 9 (counter m + 1: 4, 7, 10; blockstart
    # = counter - count = 1).  Channel 1: cells 14-17, blocks 6..7: rising edge (14-1)*8+32 = 136, falling 18*8+32 = 176, width 40 ->
    # 64, mid 156 -> [124, 188); activated in block 6 (count 2), block 7: count 3 -> part 0, counter 8: 8 - 3 = 5; block 8 (inactive 1):
    # count 4; block 9: inactive 2 > 1 -> final with the one block left, part 1, counter 10: 10 - 4 = 6.
    # In block 9 the vcm block walks the channels ONCE (:306-321): channel 0 is processed and, three blocks being buffered, emits
    # part 2 right there; only then comes channel 1's final PDU.  (SegmentDetection: the other way round, see below.)
    dict(name="vcm partial PDU of an earlier channel comes before the final PDU of a later one", delay=1, maxblocks=3, puffer=0.0,
         spec=cell_spectrum(12, [(5, 10, 2, 11, 256.0), (14, 18, 6, 7, 256.0)]),
         expect=[pdu(0, False, 0, 1, 4, 56, 120, 96), pdu(0, False, 1, 1, 7, 56, 120, 96), pdu(1, False, 0, 5, 8, 124, 188, 96),
                 pdu(0, False, 2, 1, 10, 56, 120, 96), pdu(1, True, 1, 6, 10, 124, 188, 32)]),
]

# PowerActivationChannel(N = 256, cfreq 0.5, bw 16/256, relinvovl 2, 6 dB): extract width nextpow2(16) = 16, mid 128, extract
# [120, 136) = measure range (set_startstop, :314-355); 16 - 8 = 8 samples per block; power = 16 a^2 over bins 120..135.
# thr = 10^0.6 = 3.98 (:377-381); a state change needs a ratio of 100 here.


def pac_spectrum(powers):
    s = np.ones((len(powers), N), dtype=np.float64)
    for m, p in enumerate(powers):
        s[m, 120:136] = np.sqrt(p)
    return s.astype(np.complex64)


def pac_tone_spectrum(powers, gate_bin, extra):
    """One gating bin carrying sqrt(power) (real, positive) per block, every other bin EXACTLY zero except the constant
    tones `extra` = {bin: amplitude}: the payload of an extraction is then a sum of two or three complex exponentials."""
    s = np.zeros((len(powers), N), dtype=np.complex64)
    for m, p in enumerate(powers):
        s[m, gate_bin] = np.sqrt(p)
        for b, a in extra.items():
            s[m, b] = a
    return s


def tone(k, amp, w=16):
    """What the block's extraction makes of amplitude `amp` in bin k of a w-bin slice (process_channel, :260-284): the halves of
    the windowed slice are swapped (bin k -> (k + w/2) mod w), the unnormalised backward transform follows, the first w/2 output
    samples (overlap, relinvovl 2) are dropped."""
    n = np.arange(w // 2, w)
    return amp * np.exp(2j * np.pi * ((k + w // 2) % w) * n / w)


PAC = [
    # Blocks 0..7 at 1 1 100 100 1 100 1 1.  Block 2: activate (:198-210): ID number 0, previous + current block, count 2;
    # block 3: count 3; block 4: 100/1 >= thr while active -> processed (count 4), deactivate, emit (:153-157): counter 5,
    # blockstart 5 - 4.  Block 5: 100/1 while inactive -> activated AGAIN in the very next block with ID number 1 (finished
    # channels, :308-312), previous block = block 4; block 6: deactivate: count 3, counter 7, blockstart 4.
    dict(name="PAC re-activation in the block after a deactivation", maxblocks=-1, spec=pac_spectrum([1, 1, 100, 100, 1, 100, 1, 1]),
         expect=[pdu(0, True, 0, 1, 5, 120, 136, 4 * 8), pdu(1, True, 0, 4, 7, 120, 136, 3 * 8)]),
    # Blocks at 1 1 100 100 100 100 1 1, maxblocks 0 (:162-165): every active block without a state change emits what is
    # buffered: block 3 (count 3: blocks 1, 2, 3), block 4, block 5; block 6 deactivates: final, part 3.
    dict(name="PAC maxblocks 0", maxblocks=0, spec=pac_spectrum([1, 1, 100, 100, 100, 100, 1, 1]),
         expect=[pdu(0, False, 0, 1, 4, 120, 136, 24), pdu(0, False, 1, 1, 5, 120, 136, 8), pdu(0, False, 2, 1, 6, 120, 136, 8),
                 pdu(0, True, 3, 1, 7, 120, 136, 8)]),
    # maxblocks 1: count % 1 == 0 in every block -> the same PDUs
    dict(name="PAC maxblocks 1", maxblocks=1, spec=pac_spectrum([1, 1, 100, 100, 100, 100, 1, 1]),
         expect=[pdu(0, False, 0, 1, 4, 120, 136, 24), pdu(0, False, 1, 1, 5, 120, 136, 8), pdu(0, False, 2, 1, 6, 120, 136, 8),
                 pdu(0, True, 3, 1, 7, 120, 136, 8)]),
    # maxblocks 2: count 3 (block 3): no; count 4 (block 4): 4 % 2 == 0 -> 4 blocks; count 5: no; block 6: final with 2 blocks
    dict(name="PAC maxblocks 2", maxblocks=2, spec=pac_spectrum([1, 1, 100, 100, 100, 100, 1, 1]),
         expect=[pdu(0, False, 0, 1, 5, 120, 136, 32), pdu(0, True, 1, 1, 7, 120, 136, 16)]),
]

# ---- PowerActivationChannel scenarios with their own geometry and with PAYLOAD values (entries carry "pac" = (cfreq, bw) and
# "params").  Window of the block (cr_windows, :357-375): blocklen entries e^(2 pi i phase / R), the first `rampsamps` multiplied by
# sin(pi/2 * i / (rampsamps + 1)) and MIRRORED TO THE END OF THE BLOCK-LONG TABLE (v[blocklen-1-i] = v[i]) — not to the end of the
# extraction width; process_channel multiplies bins extract_start.. by entries 0..extract_width-1 of it (:267).
#
# (1) cfreq 0.5, bw 12/256: extract width nextpow2(12) = 16, mid 128 -> [120, 136); measured [round(122), round(134)) = [122, 134);
# rampsamps = (16 - 12) / 3 = 1: entry 0 is multiplied by sin(0) = 0 (and entry 255, never used): the slice loses its FIRST bin
# and keeps its last one at full weight.  extract_start even -> deltaphase 0, phase 0 throughout.  Gate: bin 125 (slice bin 5)
# with powers 1 1 100 100 1 1 -> activation in block 2 (blocks 1, 2), block 3, deactivation in block 4: blocks 1..4, counter 5,
# blockstart 5 - 4.  Constant tones: amplitude 5 in bin 120 (slice bin 0: erased by the window), amplitude 2 in bin 135 (slice
# bin 15: untouched).  Block m contributes sqrt(P_m) * tone(5) + 2 * tone(15).
_PW = [1, 1, 100, 100, 1, 1]
PAC_GEOM = [
    dict(name="PAC window: the rising edge only, the far edge sits at the end of the block", maxblocks=-1,
         pac=(0.5, 12.0 / N), params=(120, 136, 122, 134, 8),
         spec=pac_tone_spectrum(_PW, 125, {120: 5.0, 135: 2.0}),
         expect=[dict(pdu(0, True, 0, 1, 5, 120, 136, 4 * 8),
                      payload=np.concatenate([np.sqrt(p) * tone(5, 1.0) + tone(15, 2.0) for p in _PW[1:5]]))]),
    # (2) The clamp of set_startstop (:333-336): cfreq 0.98f, bw 9/256: width nextpow2(ceil(9.0)) = 16, mid round(250.88) = 251,
    # extract_start 243, extract_stop 259 > 256 -> extract_stop = 256 and extract_start = extract_stop - BLOCKLEN = 0 (not
    # - extract_width): the block measures its power where it was told to — [round(246.38), round(255.38)) = [246, 255) — and
    # EXTRACTS BINS [0, 16).  rampsamps = ((256 - 0) - 9) / 3 = 82: slice bin k is weighted sin(pi/2 * k / 83).  The dictionary says
    # rel_cfreq = (0 + 256) / 2 / 256 = 0.5 and rel_bw = 16 / 256.  Gate in bin 250; a tone of amplitude 2 in bin 3.
    dict(name="PAC clamp at the upper band edge extracts bins 0..15", maxblocks=-1,
         pac=(0.98, 9.0 / N), params=(0, 256, 246, 255, 8),
         spec=pac_tone_spectrum(_PW, 250, {3: 2.0}),
         expect=[dict(pdu(0, True, 0, 1, 5, 0, 256, 4 * 8), rel_cfreq=0.5, rel_bw=16.0 / N,
                      payload=np.concatenate([tone(3, 2.0 * np.sin(0.5 * np.pi * 3 / 83.0))] * 4))]),
]

# SegmentDetection(ID 4, N 256, relinvovl 2, seg_start 0.5, seg_stop 1.0, 10 dB, minchandist 0.0625, puffer 0, maxblocks -1,
# delay 1).  mod_f(1.0, 1.0) = fmod(fmod(1, 1) + 1, 1) = 0 (SegmentDetection_impl.cc:700-703): stop becomes 0, start 0.5 > stop is
# swapped (:601-606): the block watches the LOWER half [0, 0.5): width (size_t)(0.5 * 256) = 128, mid (size_t)(0.25 * 256) = 64,
# start 0, stop 128, 16 cells.  Edges (:208-210): quotient index i = P[i+1] / P[i]: rising at i * 8, falling at (i + 1) * 8.
# Burst on bins 40..79 (cells 5-9): rising at 4 * 8 = 32, falling at (9 + 1) * 8 = 80: width 48 -> 64, mid 56, [24, 88).
# The counter starts at 0: activation in block 3 (count 2), block 7 still processed (count 6), emitted in block 8:
# blockend 8, blockstart 2.  A burst in the upper half (where the user asked) is not seen at all.
SD = [
    dict(name="SD mod_f(1.0) -> 0 moves the segment to the lower half", sd=(4, 0.5, 1.0), delay=1, maxblocks=-1, puffer=0.0,
         geometry=dict(start=0, stop=128, width=128, dec=8, npower=16),
         spec=cell_spectrum(12, [(5, 10, 3, 6, 256.0), (21, 26, 3, 6, 256.0)], start=0),
         expect=[pdu(0, True, None, 2, 8, 24, 88, 6 * 32)]),
    # the plain divide (:206): 256 / 0 = inf is a rising edge, 0 / 0 = NaN is nothing, 0 / 256 a falling edge
    dict(name="SD division by zero power", sd=(1, 0.125, 0.875), delay=1, maxblocks=-1, puffer=0.0,
         geometry=dict(start=32, stop=224, width=192, dec=8, npower=24),
         spec=cell_spectrum(12, _B, floor=0.0), expect=[pdu(0, True, None, 2, 8, 56, 120, 6 * 32)]),
    # Geometry by TRUNCATION (set_chan_start_stop_width_dec, :594-628; the vcm block rounds, …vcm_impl.cc:258-259).
    # seg_start 0.25f, seg_stop 0.6f = 0.60000002384: width (size_t)(0.35000002384 * 256 = 89.6) = 89, not a multiple of 8 ->
    # 89 + 7 = 96; mid (size_t)(0.5f * 0.85000002384 * 256 = 108.8) = 108 (round() would say 109 and start the segment at 61);
    # start 108 - 48 = 60, stop 156, 12 cells.  Burst on cells 3-5 (bins 84..107): rising at quotient 2 -> 2*8+60 = 76, falling at
    # quotient 5 -> (5+1)*8+60 = 108: width 32 -> 32, mid 92, [76, 108), 16 samples per block.  Counter from 0: block 3 (count 2)
    # ... block 7 (inactive 1, count 6), emitted in block 8: 8 - 6.
    dict(name="SD geometry by truncation", sd=(3, 0.25, 0.6), delay=1, maxblocks=-1, puffer=0.0,
         geometry=dict(start=60, stop=156, width=96, dec=8, npower=12),
         spec=cell_spectrum(12, [(3, 6, 3, 6, 256.0)], start=60), expect=[pdu(0, True, None, 2, 8, 76, 108, 6 * 16)]),
    # The clamp at the upper band edge (:630-633).  seg_start 0.8f, seg_stop 0.999f: width (size_t)(0.19900000095 * 256 = 50.9) = 50
    # -> 56; mid (size_t)(0.5f * 1.7990000248 * 256 = 230.27) = 230; start 230 - 28 = 202, stop 258 > 256 -> stop = 256 and
    # start = stop - BLOCKLEN = 0 (not - width): the block reports stop 256 and watches bins [0, 56).  A burst where the user asked
    # (bins 208..231) is not seen; one on cells 2-3 of the segment that IS watched (bins 16..31): rising 1*8 = 8, falling (3+1)*8 =
    # 32: width 24 -> 32, mid 20, [4, 36).
    dict(name="SD clamp at the upper band edge watches bins 0..55", sd=(2, 0.8, 0.999), delay=1, maxblocks=-1, puffer=0.0,
         geometry=dict(start=0, stop=256, width=56, dec=8, npower=7),
         spec=cell_spectrum(12, [(2, 4, 3, 6, 256.0), (26, 29, 3, 6, 256.0)], start=0), expect=[pdu(0, True, None, 2, 8, 4, 36, 6 * 16)]),
    # Partial PDUs are a pass of their own behind ALL channels (:359-362).  maxblocks 0: block 3: previous + current block buffered,
    # the pass sends both: part 0, counter 3: 3 - 2; blocks 4, 5, 6: one block each, parts 1, 2, 3, blockstart stays counter - count
    # = 1; block 7 (inactive 1): processed, part 4: 7 - 6; block 8: inactive 2 > 1: the final PDU is empty, part 5, 8 - 6; the pass
    # behind it finds nothing buffered (0 >= 0, but nothing to send: emit_unfinished_channel returns).
    dict(name="SD maxblocks 0", sd=(1, 0.125, 0.875), delay=1, maxblocks=0, puffer=0.0,
         geometry=dict(start=32, stop=224, width=192, dec=8, npower=24), spec=cell_spectrum(12, _B),
         expect=[pdu(0, False, 0, 1, 3, 56, 120, 64), pdu(0, False, 1, 1, 4, 56, 120, 32), pdu(0, False, 2, 1, 5, 56, 120, 32),
                 pdu(0, False, 3, 1, 6, 56, 120, 32), pdu(0, False, 4, 1, 7, 56, 120, 32), pdu(0, True, 5, 2, 8, 56, 120, 0)]),
    # maxblocks 1: always one block behind (two are buffered at the activation); the final PDU carries the last one
    dict(name="SD maxblocks 1", sd=(1, 0.125, 0.875), delay=1, maxblocks=1, puffer=0.0,
         geometry=dict(start=32, stop=224, width=192, dec=8, npower=24), spec=cell_spectrum(12, _B),
         expect=[pdu(0, False, 0, 1, 3, 56, 120, 32), pdu(0, False, 1, 1, 4, 56, 120, 32), pdu(0, False, 2, 1, 5, 56, 120, 32),
                 pdu(0, False, 3, 1, 6, 56, 120, 32), pdu(0, False, 4, 1, 7, 56, 120, 32), pdu(0, True, 5, 2, 8, 56, 120, 32)]),
    # maxblocks 3: block 4 (count 3): three blocks, part 0, 4 - 3; block 7 (count 6): part 1, 7 - 6; block 8: final, empty, part 2
    dict(name="SD maxblocks 3: the final PDU is empty", sd=(1, 0.125, 0.875), delay=1, maxblocks=3, puffer=0.0,
         geometry=dict(start=32, stop=224, width=192, dec=8, npower=24), spec=cell_spectrum(12, _B),
         expect=[pdu(0, False, 0, 1, 4, 56, 120, 96), pdu(0, False, 1, 1, 7, 56, 120, 96), pdu(0, True, 2, 2, 8, 56, 120, 0)]),
    # The input of "vcm partial PDU of an earlier channel comes before ..." above: in block 9 SegmentDetection first walks all
    # channels (channel 0 processed, channel 1's final PDU: counter 9, 9 - 4), THEN sends the partial PDUs (channel 0, part 2)
    dict(name="SD final PDU of a later channel comes before the partial PDU of an earlier one", sd=(1, 0.125, 0.875), delay=1,
         maxblocks=3, puffer=0.0, geometry=dict(start=32, stop=224, width=192, dec=8, npower=24),
         spec=cell_spectrum(12, [(5, 10, 2, 11, 256.0), (14, 18, 6, 7, 256.0)]),
         expect=[pdu(0, False, 0, 0, 3, 56, 120, 96), pdu(0, False, 1, 0, 6, 56, 120, 96), pdu(1, False, 0, 4, 7, 124, 188, 96),
                 pdu(1, True, 1, 5, 9, 124, 188, 32), pdu(0, False, 2, 0, 9, 56, 120, 96)]),
    # Two rising edges of INFINITE ratio (zero floor, plain divide :206): cells 3-4 at 256 -> quotient 2 = 256 / 0 = inf, rising
    # 2*8+32 = 48, quotient 4 = 0 / 256: falling (4+1)*8+32 = 72; cells 10-11 at 16 -> quotient 9 = inf, rising 104, falling
    # (11+1)*8+32 = 128; every other quotient is 0 / 0 = NaN: neither edge (:209-210).  inf > inf is false both ways: the sort
    # (:217, insertion sort at this length) leaves them in order: channel 0 = (48, 72) -> 32 bins around 60: [44, 76); channel 1 =
    # (104, 128) -> [100, 132).
    dict(name="SD two infinite rising edges keep their order", sd=(1, 0.125, 0.875), delay=1, maxblocks=-1, puffer=0.0,
         geometry=dict(start=32, stop=224, width=192, dec=8, npower=24),
         spec=cell_spectrum(12, [(3, 5, 3, 6, 256.0), (10, 12, 3, 6, 16.0)], floor=0.0),
         expect=[pdu(0, True, None, 2, 8, 44, 76, 6 * 16), pdu(1, True, None, 2, 8, 100, 132, 6 * 16)]),
    # std::upper_bound (:226) wants a falling edge strictly ABOVE the rising one: cells 4-5 at 1024, cell 6 at 1, cells 7-9 at 256:
    # rising 3*8+32 = 56 (ratio 1024) and 6*8+32 = 80 (ratio 256); falling (5+1)*8+32 = 80 and (9+1)*8+32 = 112.  (56, 80); for the
    # rising edge at 80 the falling edge AT 80 does not count: (80, 112); 80 < 80 is false, the two do not overlap (:232).
    dict(name="SD falling edge at a rising position", sd=(1, 0.125, 0.875), delay=1, maxblocks=-1, puffer=0.0,
         geometry=dict(start=32, stop=224, width=192, dec=8, npower=24),
         spec=cell_spectrum(12, [(4, 6, 3, 6, 1024.0), (7, 10, 3, 6, 256.0)]),
         expect=[pdu(0, True, None, 2, 8, 52, 84, 6 * 16), pdu(1, True, None, 2, 8, 80, 112, 6 * 16)]),
]


def check(name, got, expect):
    """got: list of (meta dict, samples) with keys chan_id, finalized, part, has_part, blockstart, blockend, vectorstart, vectorend."""
    assert len(got) == len(expect), "%s: %d PDUs, expected %d: %s" % (name, len(got), len(expect), [g[0] for g in got])
    for k, ((m, s), e) in enumerate(zip(got, expect)):
        what = "%s, PDU %d" % (name, k)
        assert int(m["chan_id"]) == e["chan_id"] and bool(m["finalized"]) == e["finalized"], (what, m)
        if e["part"] is None:
            assert not m["has_part"], (what, m)
        else:
            assert m["has_part"] and int(m["part"]) == e["part"], (what, m)
        assert (int(m["blockstart"]), int(m["blockend"])) == (e["blockstart"], e["blockend"]), (what, m)
        assert (int(m["vectorstart"]), int(m["vectorend"])) == (e["vectorstart"], e["vectorend"]), (what, m)
        assert s.size == e["nsamples"], (what, s.size)
        assert m["rel_bw"] == e.get("rel_bw", (e["vectorend"] - e["vectorstart"]) / float(N)), (what, m["rel_bw"])
        assert m["rel_cfreq"] == e.get("rel_cfreq", (e["vectorstart"] + e["vectorend"]) / 2.0 / float(N)), (what, m["rel_cfreq"])
        if "payload" in e:                          # hand-computed samples (complex exponentials): 1e-5 of the largest one
            ref = np.asarray(e["payload"], dtype=np.complex128)
            assert np.abs(np.asarray(s, dtype=np.complex128) - ref).max() <= 1e-5 * np.abs(ref).max(), (what, s[:4], ref[:4])
