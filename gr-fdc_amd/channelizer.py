"""Host-side counterpart of the reference's Python caller, python/FrequencyDomainChannelizer.py.

`FrequencyDomainChannelizer` takes the same constructor arguments as the reference hier block (:46-60),
derives the same channel parameters (:322-345) and lowers the throughput part of the flowgraph
(:200-231, :283-299) to ONE fused device pipeline behind the C-ABI (include/fdc_amd.h) instead of
4 + 6*C GNU Radio blocks.  `Pipeline` is the thin object over fdc_pipeline_* used by it, by the tests
and by bench.py.
"""
import ctypes as C
import math

import numpy as np

from . import _lib


class FREQMODE:                      # python/FrequencyDomainChannelizer.py:31-32
    normalized, basebandfs, centerfreqfs = range(3)


class VERBOSEMODE:                   # python/FrequencyDomainChannelizer.py:34-35
    NOLOG, LOGTOCONSOLE, LOGTOFILE = range(3)


class WINDOWTYPES:                   # lib/windows.h:28-32
    RECTANGULAR, HANN, RAMP = range(3)


def nextpow2(k):
    """Smallest power of two >= k (python/FrequencyDomainChannelizer.py:37-40); ValueError for k < 1."""
    if k < 1:
        raise ValueError('Cannot evaluate next power 2 of {}'.format(k))
    return 1 << int(math.ceil(math.log2(k)))


def get_opt_channelparams(blocksize, relinvovl, freq, bw):
    """(freq, bw) in INTERNAL units -> (f, l, lout, passband, stopband).

    Same decisions as the reference method (python/FrequencyDomainChannelizer.py:322-345): slice length =
    next power of two of the occupied bins with at least 20 % head-room, pass band 10 % wider than the
    signal, stop band at the slice edge unless the pass band is below 0.7, slice centred on the rounded
    carrier bin, wrapped below zero and clamped at the upper band edge."""
    occupied = blocksize * bw
    l = nextpow2(occupied)
    if l < 1.2 * occupied:
        l *= 2
    passband = float(occupied) / float(l) * 1.1
    stopband = 1.0
    if passband >= 1.0:
        passband = 1.0
    elif passband < 0.7:
        stopband = passband + 0.25
    # round(): the reference runs under Python 2 (half away from zero); Python 3's round() goes half to even
    centre = int(_round_half_away(freq * blocksize)) % blocksize
    first = centre - l / 2
    if first < 0:
        first = (first + blocksize) % blocksize
    if first + l > blocksize:
        first = blocksize - l
    return int(first), int(l), int(l) - int(l) // relinvovl, float(passband), float(stopband)


def _round_half_away(x):
    """Python 2's round() (and C's round()): to the nearest integer, ties away from zero — decided on the fraction itself,
    not on x + 0.5 (0.49999999999999994 + 0.5 rounds up to 1.0 in double; odd integers from 2^52 on would move too)."""
    ax = abs(x)
    r = math.floor(ax)
    if ax - r >= 0.5:
        r += 1.0
    return r if x >= 0 else -r


def freq_converters(freqmode, fs=1.0, centerfrequency=0.0):
    """(mode, get_freq, set_freq, get_bw, set_bw) of the three frequency conventions of the hier block
    (python/FrequencyDomainChannelizer.py:70-91): user frequency -> internal [0, 1) with DC at 0.5, and back.
    The mode may be given as the FREQMODE integer or as its name."""
    if freqmode in (FREQMODE.normalized, 'normalized'):
        return (FREQMODE.normalized, lambda f: (f + 0.5) % 1.0, lambda f: f - 0.5, lambda bw: bw % 1.0, lambda bw: bw)
    if freqmode in (FREQMODE.basebandfs, 'basebandfs'):
        return (FREQMODE.basebandfs, lambda f: (f / fs + 0.5) % 1.0, lambda f: (f - 0.5) * fs,
                lambda bw: (bw / fs) % 1.0, lambda bw: bw * fs)
    if freqmode in (FREQMODE.centerfreqfs, 'centerfreqfs'):
        return (FREQMODE.centerfreqfs, lambda f: ((f - centerfrequency) / fs + 0.5) % 1.0,
                lambda f: (f - 0.5) * fs + centerfrequency, lambda bw: (bw / fs) % 1.0, lambda bw: bw * fs)
    raise ValueError('Unknown Frequency mode. Exiting...')


def register_host(arr):
    """Pin a numpy array for fdc_pipeline_work (fdc_host_register): calls whose input / outputs are slices of pinned
    arrays are DMA'd in place.  Keep the array alive until unregister_host(arr)."""
    _lib.check(_lib.lib().fdc_host_register(arr.ctypes.data, arr.nbytes))


def unregister_host(arr):
    _lib.check(_lib.lib().fdc_host_unregister(arr.ctypes.data))


# Process-wide defaults for the kernel-choice fields of fdc_pipeline_cfg (flags, min_block_launch, host_sub_blocks) of pipelines
# created without them: how tests and A/B tools force a path.  Keys (the library itself reads no environment variable):
#   "FDC_FORCE_GENERIC" / "FDC_NO_POLY" / "FDC_NO_BLOCK" / "FDC_NO_FUSED" / "FDC_FULL_SPECTRUM" -> FDC_PIPE_* flags, "FDC_BLOCK_MIN_BLOCKS" -> min_block_launch,
#   "FDC_HOST_SUB" -> host_sub_blocks, "FDC_BLOCK_HINTS" (bit 0 nt stores, bit 1 nt loads)
defaults = {}


def _default_cfg_fields():
    flags = 0
    if defaults.get("FDC_FORCE_GENERIC"):
        flags |= _lib.FDC_PIPE_FORCE_GENERIC
    if defaults.get("FDC_NO_POLY"):
        flags |= _lib.FDC_PIPE_NO_POLY
    if defaults.get("FDC_NO_BLOCK"):
        flags |= _lib.FDC_PIPE_NO_BLOCK
    if defaults.get("FDC_FULL_SPECTRUM"):
        flags |= _lib.FDC_PIPE_FULL_SPECTRUM
    if defaults.get("FDC_WIDE_UNIFORM"):
        flags |= _lib.FDC_PIPE_WIDE_UNIFORM
    if defaults.get("FDC_NO_FUSED"):
        flags |= _lib.FDC_PIPE_NO_FUSED
    if "FDC_BLOCK_HINTS" in defaults:
        h = int(defaults["FDC_BLOCK_HINTS"])
        flags |= (0 if h & 1 else _lib.FDC_PIPE_PLAIN_STORES) | (_lib.FDC_PIPE_NT_LOADS if h & 2 else 0)
    return flags, int(defaults.get("FDC_BLOCK_MIN_BLOCKS", 0) or 0), int(defaults.get("FDC_HOST_SUB", 0) or 0)


def plan_preview(blocklen, relinvovl, channels, windowtype=WINDOWTYPES.HANN, max_blocks=64, flags=0):
    """fdc_pipeline_plan_preview: what fdc_pipeline_create would choose for this plan — (path, description, assignment) — without a device.
    assignment[c]: k >= 0 = bank k (one block-kernel launch each), -1 = the spectrum path, -2 - c0 = a copy of channel c0's output."""
    chans = [(int(f), int(l), float(p), float(s)) for (f, l, p, s) in channels]
    arr = (_lib.fdc_channel * max(1, len(chans)))()
    for i, (f, l, p, s) in enumerate(chans):
        arr[i].f, arr[i].l, arr[i].passbw, arr[i].stopbw = f, l, p, s
    cfg = _lib.fdc_pipeline_cfg(0, int(blocklen), int(relinvovl), int(windowtype), len(chans), arr, int(max_blocks), 0, 0, int(flags), 0, 0)
    buf = C.create_string_buffer(640)
    asg = (C.c_int32 * max(1, len(chans)))()
    rc = _lib.lib().fdc_pipeline_plan_preview(C.byref(cfg), buf, 640, asg)
    if rc == -1:
        raise ValueError(_lib.lib().fdc_last_error().decode())
    _lib.check(rc)
    return rc, buf.value.decode(), [int(asg[i]) for i in range(len(chans))]


class Pipeline:
    """fdc_pipeline handle: channels = [(f, l, passbw, stopbw), ...]."""

    def __init__(self, blocklen, relinvovl, channels, windowtype=WINDOWTYPES.HANN, max_blocks=64,
                 device_id=0, chunk_blocks=0, keep_spectrum=False, flags=None, min_block_launch=None, host_sub_blocks=None):
        self._h = C.c_void_p()
        self.N, self.R = int(blocklen), int(relinvovl)
        self.channels = [(int(f), int(l), float(p), float(s)) for (f, l, p, s) in channels]
        arr = (_lib.fdc_channel * max(1, len(self.channels)))()
        for i, (f, l, p, s) in enumerate(self.channels):
            arr[i].f, arr[i].l, arr[i].passbw, arr[i].stopbw = f, l, p, s
        dflags, dmin, dsub = _default_cfg_fields()
        cfg = _lib.fdc_pipeline_cfg(device_id, self.N, self.R, int(windowtype), len(self.channels), arr,
                                    int(max_blocks), int(chunk_blocks), int(bool(keep_spectrum)),
                                    dflags if flags is None else int(flags), dmin if min_block_launch is None else int(min_block_launch),
                                    dsub if host_sub_blocks is None else int(host_sub_blocks))
        rc = _lib.lib().fdc_pipeline_create(C.byref(cfg), C.byref(self._h))
        if rc == -1:
            raise ValueError(_lib.lib().fdc_last_error().decode())
        _lib.check(rc)
        self.max_blocks = int(max_blocks)
        self.keep_spectrum = bool(keep_spectrum)
        self.ovl = self.N // self.R if self.N >= self.R else 0
        self.H = self.N - self.ovl
        self.lout = [_lib.lib().fdc_pipeline_channel_lout(self._h, c) for c in range(len(self.channels))]

    # -- sizes
    def output_samples(self, nblocks):
        return int(_lib.lib().fdc_pipeline_output_samples(self._h, nblocks))

    def channel_offset(self, c, nblocks):
        return int(_lib.lib().fdc_pipeline_channel_offset(self._h, c, nblocks))

    # -- host path (what sync_block::work() would call)
    def work(self, x, want_spectrum=False, sinks=None, outs=None):
        """sinks: a gr_fdc_amd.Sinks bank fed from the device-resident spectrum of this call (needs keep_spectrum).
        outs: optional caller-owned complex64 arrays, one per channel with nblocks*lout_c samples (e.g. slices of
        buffers pinned with register_host); allocated here when None."""
        x = np.ascontiguousarray(x, dtype=np.complex64)
        if x.size % self.H:
            raise ValueError("input must be a whole number of (N - N/R)-sample items")
        nb = x.size // self.H
        if outs is None:
            outs = [np.empty(nb * lo, dtype=np.complex64) for lo in self.lout]
        else:
            if len(outs) != len(self.lout):
                raise ValueError("outs needs one array per channel")
            for o, lo in zip(outs, self.lout):
                if o.dtype != np.complex64 or not o.flags.c_contiguous or o.size != nb * lo:
                    raise ValueError("outs[c] must be contiguous complex64 with nblocks*lout_c samples")
        ptrs = (C.c_void_p * max(1, len(outs)))(*[o.ctypes.data for o in outs])
        spec = np.empty(nb * self.N, dtype=np.complex64) if want_spectrum else None
        if sinks is not None:
            _lib.check(_lib.lib().fdc_pipeline_work_sinks(self._h, x.ctypes.data, nb, ptrs,
                                                         spec.ctypes.data if spec is not None else None, sinks._h))
        else:
            _lib.check(_lib.lib().fdc_pipeline_work(self._h, x.ctypes.data, nb, ptrs,
                                                   spec.ctypes.data if spec is not None else None))
        return (outs, spec) if want_spectrum else outs

    def flush_sinks(self, sinks):
        """fdc_pipeline_flush_sinks: the pipelined form (a bank made with lookahead=True) hands out the oldest batch still inside;
        returns its block count, 0 when nothing is left.  The PDUs are then the bank's current ones (sinks.pdus())."""
        return _lib.check(_lib.lib().fdc_pipeline_flush_sinks(self._h, sinks._h))

    def sinks_latency(self, sinks):
        """Calls between an item going in and its PDUs coming out of work(..., sinks=sinks): 0 serial, 1 or 2 pipelined."""
        return int(_lib.lib().fdc_pipeline_sinks_latency(self._h, sinks._h))

    def work_sinks_raw(self, in_ptr, nblocks, out_ptrs, sinks):
        """fdc_pipeline_work_sinks on raw addresses (timing the C entry itself; the PDUs stay with the bank)."""
        return _lib.check(_lib.lib().fdc_pipeline_work_sinks(self._h, in_ptr, int(nblocks), out_ptrs, None, sinks._h))

    def work_real(self, x, want_spectrum=False):
        """Real input stream (float32 items; fdc_pipeline_work_real): the block's imaginary part is zero."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.size % self.H:
            raise ValueError("input must be a whole number of (N - N/R)-sample items")
        nb = x.size // self.H
        outs = [np.empty(nb * lo, dtype=np.complex64) for lo in self.lout]
        ptrs = (C.c_void_p * max(1, len(outs)))(*[o.ctypes.data for o in outs])
        spec = np.empty(nb * self.N, dtype=np.complex64) if want_spectrum else None
        _lib.check(_lib.lib().fdc_pipeline_work_real(self._h, x.ctypes.data, nb, ptrs,
                                                    spec.ctypes.data if spec is not None else None))
        return (outs, spec) if want_spectrum else outs

    def work_raw(self, in_ptr, nblocks, out_ptrs):
        """fdc_pipeline_work on raw addresses (out_ptrs: ctypes array of c_void_p, one per channel) — for callers that
        keep their buffers and want no per-call Python work, e.g. timing the C entry itself."""
        return _lib.check(_lib.lib().fdc_pipeline_work(self._h, in_ptr, int(nblocks), out_ptrs, None))

    def work_spectrum(self, spec_items, want_spectrum=False, sinks=None):
        """Items that are already transformed (unnormalised, fftshifted): hier block with inpveclen > 1."""
        spec_items = np.ascontiguousarray(spec_items, dtype=np.complex64)
        if spec_items.size % self.N:
            raise ValueError("input must be a whole number of N-sample spectrum items")
        nb = spec_items.size // self.N
        outs = [np.empty(nb * lo, dtype=np.complex64) for lo in self.lout]
        ptrs = (C.c_void_p * max(1, len(outs)))(*[o.ctypes.data for o in outs])
        spec = np.empty(nb * self.N, dtype=np.complex64) if want_spectrum else None
        _lib.check(_lib.lib().fdc_pipeline_work_spectrum(self._h, spec_items.ctypes.data, nb, ptrs,
                                                        spec.ctypes.data if spec is not None else None,
                                                        sinks._h if sinks is not None else None))
        return (outs, spec) if want_spectrum else outs

    def reset(self):
        _lib.lib().fdc_pipeline_reset(self._h)

    # -- device path
    def process_device(self, d_ring, first_block, nblocks, d_out, d_spectrum=None, stream=None, d_group_power=None):
        """d_group_power: device buffer of nblocks * N/16 float32 for the power of the spectrum's 16-bin groups (fdc_pipeline_process_device_power:
        what Sinks.prepare(..., from_groups=True) sums a bank's power cells from)."""
        if d_group_power:
            _lib.check(_lib.lib().fdc_pipeline_process_device_power(self._h, d_ring, int(first_block), int(nblocks), d_out, d_spectrum,
                                                                   d_group_power, stream))
        else:
            _lib.check(_lib.lib().fdc_pipeline_process_device(self._h, d_ring, int(first_block), int(nblocks), d_out,
                                                             d_spectrum, stream))

    def synchronize(self):
        _lib.check(_lib.lib().fdc_pipeline_synchronize(self._h))

    def reserve_compute_units(self, n):
        """Leave n compute units out of the persistent block kernels' grids (kernels of another stream run beside them); returns the
        number of workgroups they launch on: launch groups should hold a multiple of it in blocks."""
        return _lib.check(_lib.lib().fdc_pipeline_reserve_compute_units(self._h, int(n)))

    def stream(self):
        return _lib.lib().fdc_pipeline_stream(self._h)

    def enable_timing(self, on=True):
        _lib.check(_lib.lib().fdc_pipeline_enable_timing(self._h, int(on)))

    def last_kernel_ms(self):
        """[ms pass A, ms pass B, ms channel kernels, launch groups] summed since enable/last readout."""
        ms = (C.c_float * 4)()
        _lib.check(_lib.lib().fdc_pipeline_last_kernel_ms(self._h, ms, 4))
        return [float(v) for v in ms]

    def path(self):
        """0 generic kernels, 1 radix-16 kernels with a spectrum in memory, 2 uniform-plan two-stage path."""
        return int(_lib.lib().fdc_pipeline_path(self._h))

    def describe(self):
        """fdc_pipeline_describe: which kernels the plan was given, in words."""
        buf = C.create_string_buffer(512)
        _lib.lib().fdc_pipeline_describe(self._h, buf, 512)
        return buf.value.decode()

    def chunk_blocks(self):
        return int(_lib.lib().fdc_pipeline_chunk_blocks(self._h))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            _lib.lib().fdc_pipeline_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PipelineGroup:
    """fdc_pipeline_group handle: one work() spread over several devices (contiguous spans of the call's blocks, one per
    member, run concurrently).  devices: HIP ordinals, repeats allowed (virtual members on one GPU).  Same work() / work_real()
    as Pipeline; state (overlap history, block counter) is kept once, by the group."""

    def __init__(self, blocklen, relinvovl, channels, devices, windowtype=WINDOWTYPES.HANN, max_blocks=64, min_span_blocks=0,
                 chunk_blocks=0, keep_spectrum=False, flags=None, min_block_launch=None, host_sub_blocks=None):
        self._h = C.c_void_p()
        self.N, self.R = int(blocklen), int(relinvovl)
        self.channels = [(int(f), int(l), float(p), float(s)) for (f, l, p, s) in channels]
        self.devices = [int(d) for d in devices]
        arr = (_lib.fdc_channel * max(1, len(self.channels)))()
        for i, (f, l, p, s) in enumerate(self.channels):
            arr[i].f, arr[i].l, arr[i].passbw, arr[i].stopbw = f, l, p, s
        dflags, dmin, dsub = _default_cfg_fields()
        cfg = _lib.fdc_pipeline_cfg(0, self.N, self.R, int(windowtype), len(self.channels), arr,
                                    int(max_blocks), int(chunk_blocks), int(bool(keep_spectrum)),
                                    dflags if flags is None else int(flags), dmin if min_block_launch is None else int(min_block_launch),
                                    dsub if host_sub_blocks is None else int(host_sub_blocks))
        devs = (C.c_int32 * max(1, len(self.devices)))(*self.devices)
        rc = _lib.lib().fdc_pipeline_group_create(C.byref(cfg), devs, len(self.devices), int(min_span_blocks), C.byref(self._h))
        if rc == -1:
            raise ValueError(_lib.lib().fdc_last_error().decode())
        _lib.check(rc)
        self.max_blocks = int(max_blocks)
        self.keep_spectrum = bool(keep_spectrum)
        self.ovl = self.N // self.R if self.N >= self.R else 0
        self.H = self.N - self.ovl
        m0 = _lib.lib().fdc_pipeline_group_member(self._h, 0)
        self.lout = [_lib.lib().fdc_pipeline_channel_lout(m0, c) for c in range(len(self.channels))]

    def _run(self, fn, x, nb, want_spectrum, outs):
        if outs is None:
            outs = [np.empty(nb * lo, dtype=np.complex64) for lo in self.lout]
        ptrs = (C.c_void_p * max(1, len(outs)))(*[o.ctypes.data for o in outs])
        spec = np.empty(nb * self.N, dtype=np.complex64) if want_spectrum else None
        _lib.check(fn(self._h, x.ctypes.data, nb, ptrs, spec.ctypes.data if spec is not None else None))
        return (outs, spec) if want_spectrum else outs

    def work(self, x, want_spectrum=False, outs=None):
        x = np.ascontiguousarray(x, dtype=np.complex64)
        if x.size % self.H:
            raise ValueError("input must be a whole number of (N - N/R)-sample items")
        return self._run(_lib.lib().fdc_pipeline_group_work, x, x.size // self.H, want_spectrum, outs)

    def work_real(self, x, want_spectrum=False):
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.size % self.H:
            raise ValueError("input must be a whole number of (N - N/R)-sample items")
        return self._run(_lib.lib().fdc_pipeline_group_work_real, x, x.size // self.H, want_spectrum, None)

    def work_raw(self, in_ptr, nblocks, out_ptrs):
        return _lib.check(_lib.lib().fdc_pipeline_group_work(self._h, in_ptr, int(nblocks), out_ptrs, None))

    def reset(self):
        _lib.lib().fdc_pipeline_group_reset(self._h)

    def size(self):
        return int(_lib.lib().fdc_pipeline_group_size(self._h))

    def member_path(self, i=0):
        return int(_lib.lib().fdc_pipeline_path(_lib.lib().fdc_pipeline_group_member(self._h, i)))

    def path(self):
        return self.member_path(0)

    def member_max_blocks(self):
        return int(_lib.lib().fdc_pipeline_group_member_max_blocks(self._h))

    def last_spans(self):
        """[(first_block, nblocks)] per member for the last call (nblocks 0 = the member was idle)."""
        n = self.size()
        fb, nb = (C.c_int64 * n)(), (C.c_int32 * n)()
        _lib.check(_lib.lib().fdc_pipeline_group_last_spans(self._h, fb, nb, n))
        return [(int(fb[i]), int(nb[i])) for i in range(n)]

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            _lib.lib().fdc_pipeline_group_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FrequencyDomainChannelizer:
    """Same constructor as FDC.FrequencyDomainChannelizer (python/FrequencyDomainChannelizer.py:46-60).

    work(samples) takes a whole number of (blocksize - blocksize/relinvovl)-sample items of the input
    stream and returns the list of hier-block output ports: [spectrum (only if debug)] + one stream per
    throughput channel (:292-299, :314-315).  State (overlap history, window phase) carries over calls.
    """

    def __init__(self, inptype, inpveclen, blocksize, relinvovl,
                 throughput_channels,
                 activity_controlled_channels,
                 act_contr_threshold,
                 fs, centerfrequency, freqmode,
                 windowtype,
                 msgoutput, fileoutput, outputpath,
                 threaded,
                 activity_detection_segments, act_det_threshold, minchandist,
                 act_det_deactivation_delay, minchanflankpuffer, verbose,
                 pow_act_deactivation_delay,
                 pow_act_maxblocks, act_det_maxblocks,
                 debug, device_id=0, max_blocks=64, devices=None, pipelined=False):
        # pipelined (not an argument of the reference): the sink blocks run beside the front end of the FOLLOWING work() calls, as the
        # thread-per-block scheduler runs them beside the FFT in the reference (fdc_pipeline_work_sinks on a look-ahead bank,
        # include/fdc_amd.h): same PDUs, handed out one or two work() calls later; flush() at the end of the stream
        self.pipelined = bool(pipelined)
        self.verbose = int(verbose)
        self.itemsize = inptype
        self.debug = bool(debug)
        if self.itemsize not in (8, 4):
            raise ValueError('Unknown input type. ')            # :205-210 (there only gr_complex is reachable; the Float
        #                                                         input type of the GRC block, itemsize 4, is served here
        #                                                         by the fft_vfc front end the reference meant at :207-208)

        # frequency conventions (:70-91): everything is stored normalised to [0, 1) with DC at 0.5
        self.freqmode, self.get_freq, self.set_freq, self.get_bw, self.set_bw = freq_converters(freqmode, fs, centerfrequency)

        self.throughput_channels = self._convert(throughput_channels, self.get_channel, 'Throughput channels')
        self.activity_controlled_channels = self._convert(activity_controlled_channels, self.get_channel,
                                                          'Activity controlled channels')
        self.activity_detection_segments = self._convert(activity_detection_segments, self.get_segment,
                                                         'Activity detection segments')

        self.inpveclen = int(inpveclen) if int(inpveclen) > 0 else 1
        self.blocksize = nextpow2(blocksize)                    # :138
        self.relinvovl = nextpow2(relinvovl)                    # :139
        self.ovllen = self.blocksize // self.relinvovl          # :140
        self.inpblocklen = self.blocksize - self.ovllen         # :141
        if self.inpveclen != 1 and self.inpveclen != self.blocksize:
            raise ValueError("inpveclen must be 1 (sample stream) or blocksize (items already transformed, :284-290)")
        if self.itemsize == 4:
            # Float input is the fft_vfc front end on a sample stream; items that are already spectra are complex, and the sink
            # blocks hang on the complex chain only — refuse here, not at the first work() call
            if self.inpveclen != 1:
                raise ValueError("Float input (itemsize 4) needs inpveclen 1: pre-transformed items are complex spectra")
            if self.activity_controlled_channels or self.activity_detection_segments:
                raise ValueError("Float input (itemsize 4) cannot feed activity-controlled channels or detection segments")

        if self.verbose:                                        # runtime information, :176-193
            bar = '\n' + '#' * 32 + '\n'
            for ln in (bar, '# gr-FDC Frequency Domain Channelizer Runtime Information', bar,
                       'Blocksize     = {}'.format(self.blocksize), 'InputVecLen   = {}'.format(self.inpveclen),
                       'Relinvovl     = {}'.format(self.relinvovl), 'Ovllen        = {}'.format(self.ovllen),
                       'MsgOutput     = {}'.format(msgoutput), 'FileOutput    = {}'.format(fileoutput),
                       'Outputpath    = {}'.format(outputpath), 'Threaded      = {}'.format(threaded),
                       'Debugoutput   = {}'.format(self.debug), bar,
                       '# Throughput channels:         {}'.format(str(self.throughput_channels)),
                       '# Activity control channels:   {}'.format(str(self.activity_controlled_channels)),
                       '# Activity detection segments: {}'.format(str(self.activity_detection_segments)), bar):
                self.log(ln)
        self.channel_params = [get_opt_channelparams(self.blocksize, self.relinvovl, fr, bw)
                               for (fr, bw) in self.throughput_channels]
        if self.verbose:                                        # :223-224 (dec = blocksize / l)
            for i, (f, l, lout, pbw, sbw) in enumerate(self.channel_params):
                self.log('# Throughput Channel {}: dec={}, f={}, l={}, lout={}, bw=({}, {})'.format(
                    i, self.blocksize / l, f, l, lout, pbw, sbw))
        # activity-controlled channels (:237-251) and detection segments (:261-278) share one spectrum on the device
        self.msgoutput, self.fileoutput, self.outputpath = bool(msgoutput), bool(fileoutput), str(outputpath)
        self.sinks = None
        if self.activity_controlled_channels or self.activity_detection_segments:
            from .sinks import Sinks
            pad = int(pow_act_deactivation_delay) if int(pow_act_deactivation_delay) >= 0 else 0
            add = int(act_det_deactivation_delay) if int(act_det_deactivation_delay) >= 0 else 0
            puf = float(minchanflankpuffer) if 0.0 <= float(minchanflankpuffer) else 0.2
            self.sinks = Sinks(self.blocksize, self.relinvovl,
                               pac=[(cf, bw, i) for i, (cf, bw) in enumerate(self.activity_controlled_channels)],
                               pac_thresh=float(act_contr_threshold), pac_maxblocks=int(pow_act_maxblocks), pac_delay=pad,
                               segments=[tuple(sg) for sg in self.activity_detection_segments],
                               det_thresh=float(act_det_threshold), det_maxblocks=int(act_det_maxblocks),
                               minchandist=self.get_bw(minchandist) if self.activity_detection_segments else 0.005,
                               det_delay=add, puffer=puf, max_blocks=max_blocks, device_id=device_id,
                               det_variant=1,      # the hier block instantiates SegmentDetection (:25, :261-278)
                               verbose=self.verbose, lookahead=self.pipelined and self.inpveclen == 1)
        # devices = [ordinals]: the throughput chain of one work() call spread over several GPUs (fdc_pipeline_group); the sink
        # blocks stay on ONE device's spectrum, so a hier block with sinks keeps the single-device handle
        if devices is not None and len(devices) > 1 and self.sinks is None and self.inpveclen == 1:
            self.pipeline = PipelineGroup(self.blocksize, self.relinvovl,
                                          [(f, l, p, s) for (f, l, _lo, p, s) in self.channel_params], devices,
                                          windowtype=int(windowtype), max_blocks=max_blocks, keep_spectrum=self.debug)
        else:
            self.pipeline = Pipeline(self.blocksize, self.relinvovl,
                                     [(f, l, p, s) for (f, l, _lo, p, s) in self.channel_params],
                                     windowtype=int(windowtype), max_blocks=max_blocks,
                                     device_id=devices[0] if devices else device_id,
                                     keep_spectrum=self.debug or self.sinks is not None)
        self.N_throughput_channelizers = len(self.channel_params)
        self.messages = []          # PDUs published on "msgout" by the last work() call

    @staticmethod
    def _convert(lst, conv, what):
        out = []
        if lst is None:
            return out
        if not isinstance(lst, (list, tuple)):
            raise ValueError('{} are invalid. Exiting...'.format(what))
        for k in lst:
            c = conv(k)
            if c is None:
                raise ValueError('Cannot convert {} to channel/segment. must be list or tuple of two numbers. '.format(k))
            out.append(c)
        return out

    def get_opt_channelparams(self, freq, bw):
        return get_opt_channelparams(self.blocksize, self.relinvovl, freq, bw)

    def log(self, s):                                           # :359-371
        if self.verbose == VERBOSEMODE.LOGTOCONSOLE:
            print(str(s))
        elif self.verbose == VERBOSEMODE.LOGTOFILE:
            if not hasattr(self, 'logfile'):
                self.logfile = 'gr-FDC.FreqDomChan.log'
                with open(self.logfile, 'w') as fh:
                    fh.write('\n')
            with open(self.logfile, 'a') as fh:
                fh.write(str(s) + '\n')

    def get_channel(self, c):                                   # :349-352
        if not isinstance(c, (list, tuple)) or len(c) != 2:
            return None
        return [self.get_freq(c[0]), self.get_bw(c[1])]

    def get_segment(self, c):                                   # :354-357
        if not isinstance(c, (list, tuple)) or len(c) != 2:
            return None
        return [self.get_freq(c[0]), self.get_freq(c[1])]

    def work(self, samples):
        """Returns the hier block's stream ports; PDUs of the sink blocks ("msgout", :166-168) are left in
        self.messages as (dict, complex64 array) pairs.  Detection segments run as SegmentDetection instances, like in
        the reference hier block (:261-278)."""
        if self.inpveclen == 1 and self.itemsize == 4:
            if self.sinks is not None:
                raise ValueError("real input with sink blocks is not supported")
            res = self.pipeline.work_real(samples, want_spectrum=self.debug)
        elif self.inpveclen == 1 and self.sinks is None:
            res = self.pipeline.work(samples, want_spectrum=self.debug)
        elif self.inpveclen == 1:
            res = self.pipeline.work(samples, want_spectrum=self.debug, sinks=self.sinks)
        else:       # the front end (stream_to_vector, overlap_save, fft_vcc) is the caller's: :201, :284-290
            res = self.pipeline.work_spectrum(samples, want_spectrum=self.debug, sinks=self.sinks)
        self.messages = []
        if self.sinks is not None:
            self._publish()
        if self.debug:
            outs, spec = res
            return [spec] + outs
        return res

    def _publish(self):
        """The bank's current PDUs -> files and self.messages (appended)"""
        from .sinks import _pac_pdu, _det_pdu, _write_files
        raw = self.sinks._collect()
        pac = [_pac_pdu(m, d) for (m, d) in raw if m["kind"] == 0]
        det = [_det_pdu(m, d) for (m, d) in raw if m["kind"] == 1]
        if self.fileoutput:
            _write_files(self.outputpath, pac, True)
            _write_files(self.outputpath, det, False)
        if self.msgoutput:
            self.messages += pac + det

    def flush(self):
        """End of the stream (what the block's stop() does): the pipelined form still holds the PDUs of the last one or two work()
        calls; they are published here, batch by batch in stream order, and left in self.messages.  Serial form: nothing to do."""
        self.messages = []
        if self.sinks is not None and self.pipelined and self.inpveclen == 1:
            while self.pipeline.flush_sinks(self.sinks) > 0:
                self._publish()
        return self.messages
