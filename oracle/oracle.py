"""TEST INFRASTRUCTURE — ctypes face of oracle/libfdc_oracle.so (the CPU restatement) and, when
built, oracle/_ref/libref_windows.so (the reference's own lib/windows.h).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product
package (gr-fdc_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get("FDC_ORACLE_LIB") or os.path.join(_HERE, "libfdc_oracle.so")    # FDC_ORACLE_LIB: the sanitizer build (oracle/_san)
_REF = os.path.join(_HERE, "_ref", "libref_windows.so")

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)
_dp = C.POINTER(C.c_double)


def build(force=False):
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    if force or not os.path.exists(_LIB) or (os.path.isdir("/root/reference") and not os.path.exists(_REF)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])


def _load():
    if not os.path.exists(_LIB):
        build()
    lib = C.CDLL(_LIB)
    lib.fdco_nextpow2.restype = C.c_long
    lib.fdco_nextpow2.argtypes = [C.c_double]
    lib.fdco_channel_params.restype = C.c_int
    lib.fdco_channel_params.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, _ip, _ip, _ip, _dp, _dp]
    lib.fdco_window.restype = None
    lib.fdco_window.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.fdco_overlap_save.restype = None
    lib.fdco_overlap_save.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.fdco_vector_cut.restype = None
    lib.fdco_vector_cut.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    lib.fdco_phase_window.restype = None
    lib.fdco_phase_window.argtypes = [C.c_int, C.c_int, C.c_int, _ip, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.fdco_fft_vcc.restype = None
    lib.fdco_fft_vcc.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    lib.fdco_channelizer.restype = C.c_int
    lib.fdco_channelizer.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_int,
                                     C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int]
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def have_ref():
    return os.path.exists(_REF)


# ---- bench.py's "reference-equivalent" CPU leg (oracle/ref_equiv.c): the chain with FFTW3f + VOLK found by dlopen()
_REFEQ = os.path.join(_HERE, "libfdc_refequiv.so")
_refeq = None


def _refequiv():
    global _refeq
    if _refeq is None:
        if not os.path.exists(_REFEQ):
            subprocess.check_call(["make", "-C", _HERE, "-s", "libfdc_refequiv.so"])
        h = C.CDLL(_REFEQ)
        h.fdco_refequiv_probe.restype = C.c_int
        h.fdco_refequiv_probe.argtypes = [C.c_char_p, C.c_int]
        h.fdco_refequiv_run.restype = C.c_int
        h.fdco_refequiv_run.argtypes = [C.c_int, C.c_int, C.c_int, _ip, _ip, C.c_void_p, C.POINTER(C.c_long), C.c_void_p, C.c_int, C.c_int,
                                        C.c_double, _dp, _ip, C.c_void_p, C.c_char_p, C.c_int]
        _refeq = h
    return _refeq


def refequiv_probe():
    """None when libfftw3f and libvolk can be loaded on this host, else the reason why not."""
    err = C.create_string_buffer(256)
    return None if _refequiv().fdco_refequiv_probe(err, 256) == 0 else err.value.decode()


def refequiv_run(N, R, wintype, chans, x, fwd_threads, budget_s):
    """(Msamples/s in, passes, channel-0 output of the last pass) of the reference-equivalent chain on x (whole blocks)."""
    x = np.ascontiguousarray(x, dtype=np.complex64)
    H = N - N // R
    nb = x.size // H
    f = (C.c_int * len(chans))(*[int(c[0]) for c in chans])
    l = (C.c_int * len(chans))(*[int(c[1]) for c in chans])
    tabs, offs, off = [], [], 0
    for c in chans:
        w = window(wintype, int(c[1]), np.float32(c[2]), np.float32(c[3]), R)
        tabs.append(np.ascontiguousarray(w, dtype=np.complex64).reshape(-1))
        offs.append(off)
        off += tabs[-1].size
    wins = np.concatenate(tabs)
    woff = (C.c_long * len(chans))(*offs)
    out0 = np.empty(nb * (int(chans[0][1]) - int(chans[0][1]) // R), np.complex64)
    msps, passes = C.c_double(), C.c_int()
    err = C.create_string_buffer(256)
    rc = _refequiv().fdco_refequiv_run(N, R, len(chans), f, l, wins.ctypes.data, woff, x.ctypes.data, nb, int(fwd_threads), float(budget_s),
                                       C.byref(msps), C.byref(passes), out0.ctypes.data, err, 256)
    if rc != 0:
        raise RuntimeError(err.value.decode())
    return msps.value, passes.value, out0


def nextpow2(k):
    r = lib().fdco_nextpow2(float(k))
    if r < 0:
        raise ValueError("Cannot evaluate next power 2 of {}".format(k))
    return int(r)


def channel_params(N, R, freq, bw):
    """(f, l, lout, pbw, sbw) for an INTERNAL frequency/bandwidth pair."""
    f, l, lo = C.c_int(), C.c_int(), C.c_int()
    p, s = C.c_double(), C.c_double()
    if lib().fdco_channel_params(N, R, float(freq), float(bw), f, l, lo, p, s) != 0:
        raise ValueError("invalid channel ({}, {})".format(freq, bw))
    return f.value, l.value, lo.value, p.value, s.value


def window(wintype, blocksize, passbw, stopbw, R, step=1, normalize=False):
    w = np.empty((R, blocksize), dtype=np.complex64)
    lib().fdco_window(wintype, blocksize, passbw, stopbw, R, step, int(normalize), w.ctypes.data)
    return w


def ref_window(wintype, blocksize, passbw, stopbw, R, step=1, normalize=False):
    """The reference's own cr_win (lib/windows.h:41), compiled into oracle/_ref."""
    r = C.CDLL(_REF)
    r.ref_cr_win.restype = None
    r.ref_cr_win.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p]
    w = np.empty((R, blocksize), dtype=np.complex64)
    r.ref_cr_win(wintype, blocksize, passbw, stopbw, R, step, int(normalize), w.ctypes.data)
    return w


class OverlapSave:
    """lib/overlap_save_impl.cc (stateful, byte-level)."""

    def __init__(self, itemsize, outputlen, overlaplen):
        self.itemsize, self.outlen, self.ovl = itemsize, outputlen, overlaplen
        self.hist = np.zeros(itemsize * overlaplen, dtype=np.uint8)

    def work(self, inp):
        inp = np.ascontiguousarray(inp)
        per = self.itemsize * (self.outlen - self.ovl)
        n = inp.nbytes // per
        out = np.empty(n * self.itemsize * self.outlen, dtype=np.uint8)
        lib().fdco_overlap_save(self.itemsize, self.outlen, self.ovl, self.hist.ctypes.data,
                                inp.ctypes.data, n, out.ctypes.data)
        return out.view(inp.dtype)


def vector_cut(itemsize, veclen, offset, blocklen, inp):
    inp = np.ascontiguousarray(inp)
    n = inp.nbytes // (itemsize * veclen)
    out = np.empty(n * itemsize * blocklen, dtype=np.uint8)
    lib().fdco_vector_cut(itemsize, veclen, offset, blocklen, inp.ctypes.data, n, out.ctypes.data)
    return out.view(inp.dtype)


class PhaseWindow:
    """lib/phase_shifting_windowing_vcc_impl.cc (stateful)."""

    def __init__(self, blocklen, numphasestates, shifts, passbw, stopbw, windowtype):
        if passbw <= 0.0 or stopbw <= 0.0 or stopbw < passbw:
            raise ValueError("invalid window bandwidths")
        self.l, self.R = blocklen, numphasestates
        self.shift = ((shifts % self.R) + self.R) % self.R
        self.counter = C.c_int(0)
        self.win = window(windowtype, blocklen, passbw, stopbw, self.R, 1, False)

    def work(self, inp):
        inp = np.ascontiguousarray(inp, dtype=np.complex64)
        n = inp.size // self.l
        out = np.empty(n * self.l, dtype=np.complex64)
        lib().fdco_phase_window(self.l, self.R, self.shift, self.counter, self.win.ctypes.data,
                                inp.ctypes.data, n, out.ctypes.data)
        return out


def fft_vcc(n, forward, shift, inp):
    inp = np.ascontiguousarray(inp, dtype=np.complex64)
    out = np.empty_like(inp)
    lib().fdco_fft_vcc(n, int(forward), int(shift), inp.ctypes.data, inp.size // n, out.ctypes.data)
    return out


def channelizer(N, R, wintype, chans, x, prefix=None, first_block=0, want_spectrum=False,
                use_float=False, nthreads=1):
    """chans: list of (f, l, pbw, sbw).  Returns (list of per-channel complex64 streams, spectrum|None)."""
    H = N - N // R
    x = np.ascontiguousarray(x, dtype=np.complex64)
    nblocks = x.size // H
    Cn = len(chans)
    f = np.array([c[0] for c in chans], dtype=np.int32)
    l = np.array([c[1] for c in chans], dtype=np.int32)
    pbw = np.array([c[2] for c in chans], dtype=np.float32)
    sbw = np.array([c[3] for c in chans], dtype=np.float32)
    outs = [np.empty(nblocks * (int(li) - int(li) // R), dtype=np.complex64) for li in l]
    ptrs = (C.c_void_p * max(Cn, 1))(*[o.ctypes.data for o in outs])
    spec = np.empty(nblocks * N, dtype=np.complex64) if want_spectrum else None
    if prefix is not None:
        prefix = np.ascontiguousarray(prefix, dtype=np.complex64)
        assert prefix.size == N // R
    rc = lib().fdco_channelizer(N, R, wintype, Cn, f.ctypes.data, l.ctypes.data, pbw.ctypes.data,
                                sbw.ctypes.data, first_block,
                                prefix.ctypes.data if prefix is not None else None,
                                x.ctypes.data, nblocks, ptrs,
                                spec.ctypes.data if spec is not None else None,
                                int(use_float), nthreads)
    if rc != 0:
        raise RuntimeError("oracle channelizer failed")
    return outs, spec


# ---- stateful sinks ---------------------------------------------------------------------------------------------
class _Pdu(C.Structure):
    _fields_ = [("kind", C.c_int), ("source", C.c_int), ("chan_id", C.c_int), ("finalized", C.c_int),
                ("part", C.c_int), ("has_part", C.c_int), ("rel_bw", C.c_double), ("rel_cfreq", C.c_double),
                ("blockstart", C.c_long), ("blockend", C.c_long), ("vectorstart", C.c_long), ("vectorend", C.c_long),
                ("nsamples", C.c_long), ("samples", C.POINTER(C.c_float))]


class _PduList(C.Structure):
    _fields_ = [("pdu", C.POINTER(_Pdu)), ("n", C.c_int), ("cap", C.c_int)]


def _drain(L):
    out = []
    for i in range(L.n):
        p = L.pdu[i]
        d = dict(kind=p.kind, source=p.source, chan_id=p.chan_id, finalized=bool(p.finalized), part=p.part,
                 has_part=bool(p.has_part), rel_bw=p.rel_bw, rel_cfreq=p.rel_cfreq, blockstart=p.blockstart,
                 blockend=p.blockend, vectorstart=p.vectorstart, vectorend=p.vectorend)
        d["samples"] = np.ctypeslib.as_array(p.samples, shape=(2 * p.nsamples,)).copy().view(np.complex64) \
            if p.nsamples > 0 else np.zeros(0, np.complex64)
        out.append(d)
    lib().fdco_pdu_list_clear(C.byref(L))
    return out


def _sink_protos():
    l = lib()
    if getattr(l, "_sinks_ready", False):
        return l
    l.fdco_pdu_list_clear.argtypes = [C.POINTER(_PduList)]
    l.fdco_pac_create.restype = C.c_void_p
    l.fdco_pac_create.argtypes = [C.c_int, C.c_float, C.c_float, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
    l.fdco_pac_destroy.argtypes = [C.c_void_p]
    l.fdco_pac_params.argtypes = [C.c_void_p, _ip]
    l.fdco_pac_work.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(_PduList)]
    l.fdco_vcm_create.restype = C.c_void_p
    l.fdco_vcm_create.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_float, C.c_int, C.c_double]
    l.fdco_sd_create.restype = C.c_void_p
    l.fdco_sd_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                 C.c_int, C.c_int]
    l.fdco_vcm_destroy.argtypes = [C.c_void_p]
    l.fdco_vcm_segment_params.argtypes = [C.c_void_p, C.c_int, _ip]
    l.fdco_vcm_work.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(_PduList)]
    l._sinks_ready = True
    return l


class PowerActivationChannel:
    """lib/PowerActivationChannel_impl.cc restated (stateful)."""

    def __init__(self, blocklen, cfreq, bw, relinvovl, thresh, maxblocks, deactivation_delay, ID=0):
        self.N = blocklen
        self._h = _sink_protos().fdco_pac_create(blocklen, cfreq, bw, relinvovl, thresh, maxblocks, deactivation_delay, ID)
        if not self._h:
            raise ValueError("invalid PowerActivationChannel arguments")
        v = (C.c_int * 8)()
        lib().fdco_pac_params(self._h, v)
        (self.extract_start, self.extract_stop, self.extract_width, self.measure_start, self.measure_stop,
         self.output_len, self.output_ovl_offset, self.deltaphase) = list(v)

    def work(self, spectrum):
        spectrum = np.ascontiguousarray(spectrum, dtype=np.complex64)
        L = _PduList()
        lib().fdco_pac_work(self._h, spectrum.ctypes.data, spectrum.size // self.N, C.byref(L))
        return _drain(L)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().fdco_pac_destroy(self._h)
            self._h = None


class ActivityDetectionVcm:
    """lib/activity_detection_channelizer_vcm_impl.cc restated (stateful)."""

    def __init__(self, blocklen, segments, thresh, relinvovl, maxblocks, minchandist, deactivation_delay, puffer):
        self.N = blocklen
        seg = np.array(segments, dtype=np.float32).reshape(-1, 2)
        self._h = _sink_protos().fdco_vcm_create(blocklen, len(seg), seg.ctypes.data, thresh, relinvovl, maxblocks,
                                                 minchandist, deactivation_delay, puffer)
        if not self._h:
            raise ValueError("invalid activity_detection_channelizer_vcm arguments")
        self.segments = []
        for s in range(len(seg)):
            v = (C.c_int * 5)()
            lib().fdco_vcm_segment_params(self._h, s, v)
            self.segments.append(dict(start=v[0], stop=v[1], width=v[2], dec=v[3], npower=v[4]))

    def work(self, spectrum):
        spectrum = np.ascontiguousarray(spectrum, dtype=np.complex64)
        L = _PduList()
        lib().fdco_vcm_work(self._h, spectrum.ctypes.data, spectrum.size // self.N, C.byref(L))
        return _drain(L)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().fdco_vcm_destroy(self._h)
            self._h = None


class SegmentDetection(ActivityDetectionVcm):
    """lib/SegmentDetection_impl.cc restated (stateful): the single-segment twin of the vcm block."""

    def __init__(self, ID, blocklen, relinvovl, seg_start, seg_stop, thresh, minchandist, puffer, maxblocks, delay):
        self.N = blocklen
        self._h = _sink_protos().fdco_sd_create(ID, blocklen, relinvovl, seg_start, seg_stop, thresh, minchandist, puffer,
                                                maxblocks, delay)
        if not self._h:
            raise ValueError("invalid SegmentDetection arguments")
        v = (C.c_int * 5)()
        lib().fdco_vcm_segment_params(self._h, 0, v)
        self.segments = [dict(start=v[0], stop=v[1], width=v[2], dec=v[3], npower=v[4])]
