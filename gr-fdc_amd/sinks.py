"""Faces of the two stateful sink blocks of the reference over fdc_sinks_* (include/fdc_amd.h):

  PowerActivationChannel(blocklen, cfreq, bw, relinvovl, thresh, maxblocks, deactivation_delay, msg, fileoutput, path,
                         verbose, ID)                         — include/FDC/PowerActivationChannel.h:49
  activity_detection_channelizer_vcm(blocklen, segments, thresh, relinvovl, maxblocks, message, fileoutput, path,
                         threads, minchandist, channel_deactivation_delay, window_flank_puffer, verbose)
                                                              — include/FDC/activity_detection_channelizer_vcm.h:49

work(items) takes normalised-spectrum items and returns the PDUs the reference would publish on "msgout" as
(dict, complex64 array) pairs with the same keys (…vcm_impl.cc:415-430, PowerActivationChannel_impl.cc:222-233).
The ID strings are the reference's, timestamp of the activation included: "<Y-m-d-H-M-S>.PowActChan.<ID>.<n>(.fin|.part)",
"<Y-m-d-H-M-S>.DETECTED.<seg>.<n>" (PowerActivationChannel_impl.cc:308-312, …vcm_impl.cc:526-530); `verbose` selects the
reference's log lines (1 = stdout, 2 = its log files in the working directory).
`Sinks` is the shared-spectrum bank both faces (and the hier-block mirror) are built on.
"""
import ctypes as C
import os

import numpy as np

from . import _lib


class Sinks:
    def __init__(self, blocklen, relinvovl, pac=(), pac_thresh=6.0, pac_maxblocks=-1, pac_delay=0,
                 segments=(), det_thresh=10.0, det_maxblocks=-1, minchandist=0.005, det_delay=1, puffer=0.2,
                 max_blocks=64, device_id=0, det_variant=0, verbose=0, det_id=-1, host_decisions=False, device_payload=False,
                 threads=0, lookahead=False):
        self._h = C.c_void_p()
        self.N = int(blocklen)
        pa = (_lib.fdc_pac_cfg * max(1, len(pac)))()
        for i, (cf, bw, ident) in enumerate(pac):
            pa[i].cfreq, pa[i].bw, pa[i].id = float(cf), float(bw), int(ident)
        sg = (_lib.fdc_segment_cfg * max(1, len(segments)))()
        for i, (a, b) in enumerate(segments):
            sg[i].start, sg[i].stop = float(a), float(b)
        cfg = _lib.fdc_sinks_cfg(device_id, self.N, int(relinvovl), len(pac), pa, float(pac_thresh), int(pac_maxblocks),
                                 int(pac_delay), len(segments), sg, float(det_thresh), int(det_maxblocks),
                                 float(minchandist), int(det_delay), float(puffer), int(max_blocks), int(det_variant),
                                 int(verbose), int(det_id),
                                 (_lib.FDC_SINKS_HOST_DECISIONS if host_decisions else 0) |
                                 (_lib.FDC_SINKS_DEVICE_PAYLOAD if device_payload else 0) |
                                 (_lib.FDC_SINKS_LOOKAHEAD if lookahead else 0), int(threads))
        self.device_payload = bool(device_payload)
        rc = _lib.lib().fdc_sinks_create(C.byref(cfg), C.byref(self._h))
        if rc == -1:
            raise ValueError(_lib.lib().fdc_last_error().decode())
        _lib.check(rc)
        self.max_blocks = int(max_blocks)
        self.npac, self.nseg = len(pac), len(segments)

    def pac_params(self, i):
        v = (C.c_int32 * 8)()
        _lib.check(_lib.lib().fdc_sinks_pac_params(self._h, i, v))
        return dict(zip(("extract_start", "extract_stop", "extract_width", "measure_start", "measure_stop",
                         "output_len", "output_ovl_offset", "deltaphase"), list(v)))

    def segment_params(self, i):
        v = (C.c_int32 * 5)()
        _lib.check(_lib.lib().fdc_sinks_segment_params(self._h, i, v))
        return dict(zip(("start", "stop", "width", "dec", "npower"), list(v)))

    def spectrum_ptr(self):
        """Where the items of the batch the NEXT submit / work call reads go (with lookahead=True: alternates, ask per batch)."""
        return _lib.lib().fdc_sinks_spectrum(self._h)

    def spectrum_ahead_ptr(self):
        """lookahead=True: where the items of the batch AFTER the next submit go; to be filled on fill_stream() (include/fdc_amd.h)."""
        return _lib.lib().fdc_sinks_spectrum_ahead(self._h)

    def stream(self):
        return _lib.lib().fdc_sinks_stream(self._h)

    def fill_stream(self):
        return _lib.lib().fdc_sinks_fill_stream(self._h)

    def group_power_ptr(self):
        """Device buffer (max_blocks * N/16 float32) for the group powers that go with spectrum_ptr(): Pipeline.process_device(..., d_group_power=)."""
        return _lib.lib().fdc_sinks_group_power(self._h)

    def group_power_ahead_ptr(self):
        return _lib.lib().fdc_sinks_group_power_ahead(self._h)

    def prepare(self, nblocks, ahead=True, from_groups=False):
        """Power cells of the batch in spectrum_ahead_ptr() (or spectrum_ptr()) on the fill stream, behind whatever filled it.
        from_groups: summed from the group powers the forward kernel left in group_power(_ahead)_ptr() instead of from the spectrum
        (ahead=False then also works on a bank without lookahead, on its own stream)."""
        if from_groups:
            _lib.check(_lib.lib().fdc_sinks_prepare_from_groups(self._h, int(nblocks), 1 if ahead else 0))
        else:
            _lib.check(_lib.lib().fdc_sinks_prepare(self._h, int(nblocks), 1 if ahead else 0))

    def engine(self):
        """1 = decisions on the device (default), 0 = on host threads (verbose != 0, host_decisions, very fine segments)"""
        return int(_lib.lib().fdc_sinks_engine(self._h))

    def _collect(self):
        n = _lib.lib().fdc_sinks_pdu_count(self._h)
        if n <= 0:
            return []
        arr = (_lib.fdc_pdu * n)()
        _lib.check(_lib.lib().fdc_sinks_pdus(self._h, arr, n))
        if self.device_payload and self.engine() == 1:      # payloads stay on the device: (address, sample count) instead of arrays
            return [(dict(id=p.id.decode(), kind=p.kind, source=p.source, chan_id=p.chan_id, finalized=bool(p.finalized), part=p.part,
                          has_part=bool(p.has_part), rel_bw=p.rel_bw, rel_cfreq=p.rel_cfreq, blockstart=p.blockstart,
                          blockend=p.blockend, vectorstart=p.vectorstart, vectorend=p.vectorend), (p.samples or 0, p.nsamples))
                    for p in arr]
        # payloads that sit one behind the other in the handle's buffer are copied out as ONE array and sliced
        out, i = [], 0
        while i < n:
            j, base, end = i, arr[i].samples, (arr[i].samples or 0) + 8 * arr[i].nsamples
            while j + 1 < n and arr[j + 1].samples == end and arr[j + 1].nsamples > 0:
                j += 1
                end += 8 * arr[j].nsamples
            total = (end - (base or 0)) // 8
            if total > 0:
                blob = np.frombuffer((C.c_float * (2 * total)).from_address(base), dtype=np.complex64).copy()
            else:
                blob = np.zeros(0, np.complex64)
            off = 0
            for k in range(i, j + 1):
                p = arr[k]
                out.append((dict(id=p.id.decode(), kind=p.kind, source=p.source, chan_id=p.chan_id, finalized=bool(p.finalized), part=p.part,
                                 has_part=bool(p.has_part), rel_bw=p.rel_bw, rel_cfreq=p.rel_cfreq,
                                 blockstart=p.blockstart, blockend=p.blockend, vectorstart=p.vectorstart,
                                 vectorend=p.vectorend), blob[off:off + p.nsamples]))
                off += p.nsamples
            i = j + 1
        return out

    def pdus(self):
        """PDUs of the last work call (also of a fdc_pipeline_work_sinks call that fed this bank) as (meta, samples)."""
        return self._collect()

    def work(self, spectrum):
        spectrum = np.ascontiguousarray(spectrum, dtype=np.complex64).reshape(-1)     # items one behind the other
        if spectrum.size % self.N:
            raise ValueError("input is not a whole number of spectrum items")
        n = spectrum.size // self.N
        out = []
        for a in range(0, n, self.max_blocks):          # batches of at most max_blocks items
            b = min(n, a + self.max_blocks)
            _lib.check(_lib.lib().fdc_sinks_work(self._h, spectrum[a * self.N:].ctypes.data, b - a))
            out += self._collect()
        return out

    def work_device(self, nblocks):
        _lib.check(_lib.lib().fdc_sinks_work_device(self._h, int(nblocks)))
        return self._collect()

    def submit_device(self, nblocks):
        """Two-deep form (fdc_sinks_submit_device): enqueue the batch in the spectrum buffer; returns the PDUs of the batch
        submitted BEFORE it (empty list for the first one)."""
        done = _lib.check(_lib.lib().fdc_sinks_submit_device(self._h, int(nblocks)))
        return self._collect() if done > 0 else []

    def flush(self):
        done = _lib.check(_lib.lib().fdc_sinks_flush(self._h))
        return self._collect() if done > 0 else []

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            _lib.lib().fdc_sinks_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SinksGroup(Sinks):
    """fdc_sinks_group: the bank cut by frequency band over several devices (devices: HIP ordinals, repeats allowed).  Member i holds
    a run of the PowerActivationChannels and a run of the segments (bank order, equal counts) and copies only the bins that run reads;
    the PDUs come back merged into the order one bank emits them in.  work() as for Sinks."""

    def __init__(self, blocklen, relinvovl, devices, pac=(), pac_thresh=6.0, pac_maxblocks=-1, pac_delay=0,
                 segments=(), det_thresh=10.0, det_maxblocks=-1, minchandist=0.005, det_delay=1, puffer=0.2,
                 max_blocks=64, det_variant=0, verbose=0, det_id=-1, host_decisions=False, threads=0):
        self._h = C.c_void_p()
        self._g = C.c_void_p()
        self.N = int(blocklen)
        pa = (_lib.fdc_pac_cfg * max(1, len(pac)))()
        for i, (cf, bw, ident) in enumerate(pac):
            pa[i].cfreq, pa[i].bw, pa[i].id = float(cf), float(bw), int(ident)
        sg = (_lib.fdc_segment_cfg * max(1, len(segments)))()
        for i, (a, b) in enumerate(segments):
            sg[i].start, sg[i].stop = float(a), float(b)
        cfg = _lib.fdc_sinks_cfg(0, self.N, int(relinvovl), len(pac), pa, float(pac_thresh), int(pac_maxblocks),
                                 int(pac_delay), len(segments), sg, float(det_thresh), int(det_maxblocks),
                                 float(minchandist), int(det_delay), float(puffer), int(max_blocks), int(det_variant),
                                 int(verbose), int(det_id), _lib.FDC_SINKS_HOST_DECISIONS if host_decisions else 0, int(threads), 0)
        devs = (C.c_int32 * max(1, len(devices)))(*[int(d) for d in devices])
        rc = _lib.lib().fdc_sinks_group_create(C.byref(cfg), devs, len(devices), C.byref(self._g))
        if rc == -1:
            raise ValueError(_lib.lib().fdc_last_error().decode())
        _lib.check(rc)
        self.device_payload = False
        self.max_blocks = int(max_blocks)
        self.npac, self.nseg = len(pac), len(segments)

    def engine(self):
        return 1

    def size(self):
        return int(_lib.lib().fdc_sinks_group_size(self._g))

    def members(self):
        """[(device, band_lo, band_hi, npac, nseg)] per member"""
        out = []
        for i in range(self.size()):
            v = [C.c_int32() for _ in range(5)]
            _lib.check(_lib.lib().fdc_sinks_group_member_info(self._g, i, *[C.byref(x) for x in v]))
            out.append(tuple(int(x.value) for x in v))
        return out

    def _pdu_array(self):
        n = _lib.lib().fdc_sinks_group_pdu_count(self._g)
        if n <= 0:
            return None, 0
        arr = (_lib.fdc_pdu * n)()
        _lib.check(_lib.lib().fdc_sinks_group_pdus(self._g, arr, n))
        return arr, n

    def _collect(self):
        arr, n = self._pdu_array()
        out = []
        for k in range(n):
            p = arr[k]
            data = np.frombuffer((C.c_float * (2 * p.nsamples)).from_address(p.samples), dtype=np.complex64).copy() if p.nsamples > 0 \
                else np.zeros(0, np.complex64)
            out.append((dict(id=p.id.decode(), kind=p.kind, source=p.source, chan_id=p.chan_id, finalized=bool(p.finalized), part=p.part,
                             has_part=bool(p.has_part), rel_bw=p.rel_bw, rel_cfreq=p.rel_cfreq, blockstart=p.blockstart,
                             blockend=p.blockend, vectorstart=p.vectorstart, vectorend=p.vectorend), data))
        return out

    def work(self, spectrum):
        spectrum = np.ascontiguousarray(spectrum, dtype=np.complex64).reshape(-1)
        if spectrum.size % self.N:
            raise ValueError("input is not a whole number of spectrum items")
        n = spectrum.size // self.N
        out = []
        for a in range(0, n, self.max_blocks):
            b = min(n, a + self.max_blocks)
            _lib.check(_lib.lib().fdc_sinks_group_work(self._g, spectrum[a * self.N:].ctypes.data, b - a))
            out += self._collect()
        return out

    def close(self):
        if getattr(self, "_g", None) is not None and self._g:
            _lib.lib().fdc_sinks_group_destroy(self._g)
            self._g = C.c_void_p()


def _pac_pdu(meta, data):
    """dict keys of PowerActivationChannel_impl.cc:222-230"""
    d = {"ID": meta["id"] + (".fin" if meta["finalized"] else ".part"),
         "finalized": meta["finalized"], "part": meta["part"], "rel_cfreq": meta["rel_cfreq"], "rel_bw": meta["rel_bw"],
         "blockstart": meta["blockstart"], "blockend": meta["blockend"]}
    return d, data


def _det_pdu(meta, data):
    """dict keys of activity_detection_channelizer_vcm_impl.cc:415-427 / :472-483"""
    d = {"ID": meta["id"], "finalized": meta["finalized"]}
    if meta["has_part"]:
        d["part"] = meta["part"]
    d.update(rel_bw=meta["rel_bw"], rel_cfreq=meta["rel_cfreq"], blockstart=meta["blockstart"],
             blockend=meta["blockend"], vectorstart=meta["vectorstart"], vectorend=meta["vectorend"])
    return d, data


def _write_files(path, pdus, pac):
    """<path>/<ID>.fin and <path>/<ID>.parted.<n>, raw complex64 (…vcm_impl.cc:431-439,488-496;
    PowerActivationChannel_impl.cc:235-244)."""
    for d, data in pdus:
        base = d["ID"]
        if pac:
            base = base.rsplit(".", 1)[0]
        name = base + (".fin" if d["finalized"] else ".parted.%d" % d.get("part", 0))
        try:
            data.tofile(os.path.join(path, name))
        except OSError as e:                       # the reference prints to cerr and carries on
            print("Cannot write to file", name, e)


class PowerActivationChannel:
    def __init__(self, blocklen, cfreq, bw, relinvovl, thresh, maxblocks, deactivation_delay, msg, fileoutput, path,
                 verbose, ID, device_id=0, max_blocks=64):
        self.msg, self.fileoutput, self.path = bool(msg), bool(fileoutput), str(path)
        self.bank = Sinks(blocklen, relinvovl, pac=[(cfreq, bw, ID)], pac_thresh=thresh, pac_maxblocks=maxblocks,
                          pac_delay=deactivation_delay, max_blocks=max_blocks, device_id=device_id, verbose=verbose)
        self.params = self.bank.pac_params(0)

    def work(self, spectrum):
        pdus = [_pac_pdu(m, d) for (m, d) in self.bank.work(spectrum)]
        if self.fileoutput:
            _write_files(self.path, pdus, True)
        return pdus


class activity_detection_channelizer_vcm:
    def __init__(self, blocklen, segments, thresh, relinvovl, maxblocks, message, fileoutput, path, threads,
                 minchandist, channel_deactivation_delay, window_flank_puffer, verbose, device_id=0, max_blocks=64):
        self.msg, self.fileoutput, self.path = bool(message), bool(fileoutput), str(path)
        for s in segments:
            if len(s) != 2:
                raise ValueError("Segment is incorrect. must be of size 2")
        self.bank = Sinks(blocklen, relinvovl, segments=[tuple(s) for s in segments], det_thresh=thresh,
                          det_maxblocks=maxblocks, minchandist=minchandist, det_delay=channel_deactivation_delay,
                          puffer=window_flank_puffer, max_blocks=max_blocks, device_id=device_id, verbose=verbose)
        self.segments = [self.bank.segment_params(i) for i in range(len(segments))]

    def work(self, spectrum):
        pdus = [_det_pdu(m, d) for (m, d) in self.bank.work(spectrum)]
        if self.fileoutput:
            _write_files(self.path, pdus, False)
        return pdus


class SegmentDetection:
    """gr::FDC::SegmentDetection::make(ID, blocklen, relinvovl, seg_start, seg_stop, thresh, minchandist,
    window_flank_puffer, maxblocks_to_emit, channel_deactivation_delay, messageoutput, fileoutput, path, threads, verbose)
    — include/FDC/SegmentDetection.h:49.  The single-segment twin of the vcm block that the hier block instantiates."""

    def __init__(self, ID, blocklen, relinvovl, seg_start, seg_stop, thresh, minchandist, window_flank_puffer,
                 maxblocks_to_emit, channel_deactivation_delay, messageoutput, fileoutput, path, threads, verbose,
                 device_id=0, max_blocks=64):
        self.ID, self.msg, self.fileoutput, self.path = int(ID), bool(messageoutput), bool(fileoutput), str(path)
        self.bank = Sinks(blocklen, relinvovl, segments=[(seg_start, seg_stop)], det_thresh=thresh,
                          det_maxblocks=maxblocks_to_emit, minchandist=minchandist, det_delay=channel_deactivation_delay,
                          puffer=window_flank_puffer, max_blocks=max_blocks, device_id=device_id, det_variant=1,
                          verbose=verbose, det_id=self.ID)
        self.segment = self.bank.segment_params(0)

    def work(self, spectrum):
        pdus = []
        for (m, d) in self.bank.work(spectrum):
            pdus.append(_det_pdu(dict(m, source=self.ID), d))
        if self.fileoutput:
            _write_files(self.path, pdus, False)
        return pdus
