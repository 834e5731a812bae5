"""Single-block faces with the reference's factory signatures (swig/FDC_swig.i:20-32 exposes
FDC.overlap_save(...), FDC.vector_cut_vxx(...), FDC.phase_shifting_windowing_vcc(...)).

GNU Radio's scheduler is not part of this package: each face offers work(input) -> output on numpy
arrays, the same contract sync_block::work() has (n input items in, n output items out), and runs
on the GPU through the C-ABI.
"""
import ctypes as C

import numpy as np

from . import _lib


class _Block:
    _destroy = None

    def __init__(self):
        self._h = C.c_void_p()

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            getattr(_lib.lib(), self._destroy)(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class overlap_save(_Block):
    """gr::FDC::overlap_save::make(itemsize, outputlen, overlaplen) — include/FDC/overlap_save.h:49."""
    _destroy = "fdc_overlap_save_destroy"

    def __init__(self, itemsize, outputlen, overlaplen, device_id=0):
        super().__init__()
        self.itemsize, self.outputlen, self.overlaplen = int(itemsize), int(outputlen), int(overlaplen)
        _lib.check(_lib.lib().fdc_overlap_save_create(device_id, self.itemsize, self.outputlen, self.overlaplen,
                                                     C.byref(self._h)))

    def work(self, inp):
        inp = np.ascontiguousarray(inp)
        per = self.itemsize * (self.outputlen - self.overlaplen)
        if inp.nbytes % per:
            raise ValueError("input is not a whole number of items")
        n = inp.nbytes // per
        out = np.empty(n * self.itemsize * self.outputlen, dtype=np.uint8)
        _lib.check(_lib.lib().fdc_overlap_save_work(self._h, inp.ctypes.data, n, out.ctypes.data))
        return out.view(inp.dtype) if (out.nbytes % inp.dtype.itemsize) == 0 else out


class vector_cut_vxx(_Block):
    """gr::FDC::vector_cut_vxx::make(itemsize, veclen, offset, blocklen) — include/FDC/vector_cut_vxx.h:49."""
    _destroy = "fdc_vector_cut_destroy"

    def __init__(self, itemsize, veclen, offset, blocklen, device_id=0):
        super().__init__()
        self.itemsize, self.veclen, self.offset, self.blocklen = int(itemsize), int(veclen), int(offset), int(blocklen)
        _lib.check(_lib.lib().fdc_vector_cut_create(device_id, self.itemsize, self.veclen, self.offset, self.blocklen,
                                                   C.byref(self._h)))

    def work(self, inp):
        inp = np.ascontiguousarray(inp)
        per = self.itemsize * self.veclen
        if inp.nbytes % per:
            raise ValueError("input is not a whole number of items")
        n = inp.nbytes // per
        out = np.empty(n * self.itemsize * self.blocklen, dtype=np.uint8)
        _lib.check(_lib.lib().fdc_vector_cut_work(self._h, inp.ctypes.data, n, out.ctypes.data))
        return out.view(inp.dtype) if (out.nbytes % inp.dtype.itemsize) == 0 else out


class phase_shifting_windowing_vcc(_Block):
    """gr::FDC::phase_shifting_windowing_vcc::make(blocklen, numphasestates, shifts, passbw, stopbw, windowtype)
    — include/FDC/phase_shifting_windowing_vcc.h:49.  Raises ValueError where the reference constructor
    throws std::invalid_argument (lib/phase_shifting_windowing_vcc_impl.cc:46-53)."""
    _destroy = "fdc_phase_window_destroy"

    def __init__(self, blocklen, numphasestates, shifts, passbw, stopbw, windowtype, device_id=0):
        super().__init__()
        self.blocklen = int(blocklen)
        rc = _lib.lib().fdc_phase_window_create(device_id, self.blocklen, int(numphasestates), int(shifts),
                                                float(passbw), float(stopbw), int(windowtype), C.byref(self._h))
        if rc == -1:
            raise ValueError(_lib.lib().fdc_last_error().decode())
        _lib.check(rc)

    def work(self, inp):
        inp = np.ascontiguousarray(inp, dtype=np.complex64)
        if inp.size % self.blocklen:
            raise ValueError("input is not a whole number of items")
        out = np.empty_like(inp)
        _lib.check(_lib.lib().fdc_phase_window_work(self._h, inp.ctypes.data, inp.size // self.blocklen,
                                                   out.ctypes.data))
        return out


def fft_vcc(n, forward, shift, inp, device_id=0):
    """gr-fft fft_vcc(n, forward, rectangular window, shift) on the GPU (py:206,228 call sites)."""
    inp = np.ascontiguousarray(inp, dtype=np.complex64)
    if inp.size % n:
        raise ValueError("input is not a whole number of items")
    out = np.empty_like(inp)
    _lib.check(_lib.lib().fdc_fft_vcc(device_id, int(n), int(bool(forward)), int(bool(shift)), inp.ctypes.data,
                                     inp.size // n, out.ctypes.data))
    return out


def window_table(windowtype, blocklen, passbw, stopbw, numphasestates, step=1, normalize=False):
    """The [numphasestates][blocklen] complex64 table a phase_shifting_windowing_vcc instance multiplies by
    (lib/windows.h cr_win).  Host-side design; no device needed."""
    w = np.empty((numphasestates, blocklen), dtype=np.complex64)
    _lib.check(_lib.lib().fdc_window_table(int(windowtype), int(blocklen), float(passbw), float(stopbw),
                                          int(numphasestates), int(step), int(bool(normalize)), w.ctypes.data))
    return w
