"""ctypes binding of libfdc_amd.so (the C-ABI in include/fdc_amd.h).

There is no CPU fallback: if the library is missing or no HIP device is visible the calls raise.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FDC_AMD_LIB") or os.path.join(_HERE, "libfdc_amd.so")    # FDC_AMD_LIB: A/B testing of builds

FDC_OK = 0
STATUS_NAMES = {0: "FDC_OK", -1: "FDC_ERR_INVALID_ARGUMENT", -2: "FDC_ERR_HIP", -3: "FDC_ERR_NO_DEVICE",
                -4: "FDC_ERR_UNSUPPORTED", -5: "FDC_ERR_NOMEM"}


class FdcError(RuntimeError):
    def __init__(self, status, text):
        self.status = status
        super().__init__("%s: %s" % (STATUS_NAMES.get(status, status), text))


class fdc_channel(C.Structure):
    _fields_ = [("f", C.c_int32), ("l", C.c_int32), ("passbw", C.c_float), ("stopbw", C.c_float)]


class fdc_pipeline_cfg(C.Structure):
    _fields_ = [("device_id", C.c_int32), ("blocklen", C.c_int32), ("relinvovl", C.c_int32),
                ("windowtype", C.c_int32), ("nchannels", C.c_int32), ("channels", C.POINTER(fdc_channel)),
                ("max_blocks", C.c_int32), ("chunk_blocks", C.c_int32), ("keep_spectrum", C.c_int32),
                ("flags", C.c_int32), ("min_block_launch", C.c_int32), ("host_sub_blocks", C.c_int32)]


FDC_PIPE_FORCE_GENERIC, FDC_PIPE_NO_POLY, FDC_PIPE_NO_BLOCK, FDC_PIPE_PLAIN_STORES, FDC_PIPE_NT_LOADS = 1, 2, 4, 8, 16
FDC_PIPE_FULL_SPECTRUM = 32
FDC_PIPE_WIDE_UNIFORM = 64
FDC_PIPE_NO_FUSED = 128


class fdc_pac_cfg(C.Structure):
    _fields_ = [("cfreq", C.c_float), ("bw", C.c_float), ("id", C.c_int32)]


class fdc_segment_cfg(C.Structure):
    _fields_ = [("start", C.c_float), ("stop", C.c_float)]


class fdc_sinks_cfg(C.Structure):
    _fields_ = [("device_id", C.c_int32), ("blocklen", C.c_int32), ("relinvovl", C.c_int32),
                ("npac", C.c_int32), ("pac", C.POINTER(fdc_pac_cfg)),
                ("pac_thresh_db", C.c_float), ("pac_maxblocks", C.c_int32), ("pac_deactivation_delay", C.c_int32),
                ("nseg", C.c_int32), ("seg", C.POINTER(fdc_segment_cfg)),
                ("det_thresh_db", C.c_float), ("det_maxblocks", C.c_int32), ("minchandist", C.c_float),
                ("det_deactivation_delay", C.c_int32), ("window_flank_puffer", C.c_double), ("max_blocks", C.c_int32),
                ("det_variant", C.c_int32), ("verbose", C.c_int32), ("det_id", C.c_int32), ("flags", C.c_int32),
                ("threads", C.c_int32), ("seg_id_base", C.c_int32)]


FDC_SINKS_HOST_DECISIONS = 1
FDC_SINKS_DEVICE_PAYLOAD = 2
FDC_SINKS_LOOKAHEAD = 4


class fdc_pdu(C.Structure):
    _fields_ = [("kind", C.c_int32), ("source", C.c_int32), ("chan_id", C.c_int32), ("finalized", C.c_int32),
                ("part", C.c_int32), ("has_part", C.c_int32), ("rel_bw", C.c_double), ("rel_cfreq", C.c_double),
                ("blockstart", C.c_int64), ("blockend", C.c_int64), ("vectorstart", C.c_int64),
                ("vectorend", C.c_int64), ("nsamples", C.c_int64), ("samples", C.c_void_p), ("id", C.c_char * 72)]


# every symbol include/fdc_amd.h declares: (restype, argtypes)
_vp = C.c_void_p
SYMBOLS = {
    "fdc_last_error": (C.c_char_p, []),
    "fdc_version": (C.c_char_p, []),
    "fdc_device_count": (C.c_int, []),
    "fdc_selftest_devices": (C.c_int, []),
    "fdc_selftest_exception_barrier": (C.c_int, []),
    "fdc_pipeline_create": (C.c_int, [C.POINTER(fdc_pipeline_cfg), C.POINTER(_vp)]),
    "fdc_pipeline_destroy": (None, [_vp]),
    "fdc_pipeline_input_samples": (C.c_int64, [_vp, C.c_int]),
    "fdc_pipeline_output_samples": (C.c_int64, [_vp, C.c_int]),
    "fdc_pipeline_channel_offset": (C.c_int64, [_vp, C.c_int, C.c_int]),
    "fdc_pipeline_channel_lout": (C.c_int32, [_vp, C.c_int]),
    "fdc_pipeline_work": (C.c_int, [_vp, _vp, C.c_int, C.POINTER(_vp), _vp]),
    "fdc_pipeline_work_real": (C.c_int, [_vp, _vp, C.c_int, C.POINTER(_vp), _vp]),
    "fdc_pipeline_work_span": (C.c_int, [_vp, _vp, _vp, C.c_int64, C.c_int, C.POINTER(_vp), _vp]),
    "fdc_pipeline_work_span_real": (C.c_int, [_vp, _vp, _vp, C.c_int64, C.c_int, C.POINTER(_vp), _vp]),
    "fdc_pipeline_group_create": (C.c_int, [C.POINTER(fdc_pipeline_cfg), C.POINTER(C.c_int32), C.c_int, C.c_int, C.POINTER(_vp)]),
    "fdc_pipeline_group_destroy": (None, [_vp]),
    "fdc_pipeline_group_work": (C.c_int, [_vp, _vp, C.c_int, C.POINTER(_vp), _vp]),
    "fdc_pipeline_group_work_real": (C.c_int, [_vp, _vp, C.c_int, C.POINTER(_vp), _vp]),
    "fdc_pipeline_group_reset": (None, [_vp]),
    "fdc_pipeline_group_size": (C.c_int32, [_vp]),
    "fdc_pipeline_group_member": (_vp, [_vp, C.c_int]),
    "fdc_pipeline_group_device": (C.c_int32, [_vp, C.c_int]),
    "fdc_pipeline_group_member_max_blocks": (C.c_int32, [_vp]),
    "fdc_pipeline_group_last_spans": (C.c_int, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.c_int]),
    "fdc_host_register": (C.c_int, [_vp, C.c_size_t]),
    "fdc_host_unregister": (C.c_int, [_vp]),
    "fdc_pipeline_reset": (None, [_vp]),
    "fdc_pipeline_process_device": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp, _vp, _vp]),
    "fdc_pipeline_synchronize": (C.c_int, [_vp]),
    "fdc_pipeline_stream": (_vp, [_vp]),
    "fdc_pipeline_reserve_compute_units": (C.c_int, [_vp, C.c_int32]),
    "fdc_pipeline_chunk_blocks": (C.c_int32, [_vp]),
    "fdc_pipeline_path": (C.c_int32, [_vp]),
    "fdc_pipeline_describe": (C.c_int32, [_vp, C.c_char_p, C.c_int32]),
    "fdc_pipeline_plan_preview": (C.c_int, [C.POINTER(fdc_pipeline_cfg), C.c_char_p, C.c_int32, C.POINTER(C.c_int32)]),
    "fdc_pipeline_enable_timing": (C.c_int, [_vp, C.c_int]),
    "fdc_pipeline_last_kernel_ms": (C.c_int, [_vp, C.POINTER(C.c_float), C.c_int]),
    "fdc_sinks_create": (C.c_int, [C.POINTER(fdc_sinks_cfg), C.POINTER(_vp)]),
    "fdc_sinks_destroy": (None, [_vp]),
    "fdc_pipeline_work_spectrum": (C.c_int, [_vp, _vp, C.c_int, C.POINTER(_vp), _vp, _vp]),
    "fdc_pipeline_work_sinks": (C.c_int, [_vp, _vp, C.c_int, C.POINTER(_vp), _vp, _vp]),
    "fdc_device_numa_node": (C.c_int, [C.c_int]),
    "fdc_selftest_worker_placement": (C.c_int, [C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "fdc_pipeline_process_device_power": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp, _vp, _vp, _vp]),
    "fdc_sinks_group_power": (_vp, [_vp]),
    "fdc_sinks_group_power_ahead": (_vp, [_vp]),
    "fdc_sinks_prepare_from_groups": (C.c_int, [_vp, C.c_int, C.c_int]),
    "fdc_pipeline_flush_sinks": (C.c_int, [_vp, _vp]),
    "fdc_pipeline_sinks_latency": (C.c_int32, [_vp, _vp]),
    "fdc_sinks_work": (C.c_int, [_vp, _vp, C.c_int]),
    "fdc_sinks_work_band": (C.c_int, [_vp, _vp, C.c_int, C.c_int32, C.c_int32]),
    "fdc_sinks_read_band": (C.c_int, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "fdc_sinks_pdu_emit_items": (C.c_int, [_vp, C.POINTER(C.c_int32), C.c_int]),
    "fdc_sinks_pdu_emit_order": (C.c_int, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int]),
    "fdc_sinks_group_create": (C.c_int, [C.POINTER(fdc_sinks_cfg), C.POINTER(C.c_int32), C.c_int, C.POINTER(_vp)]),
    "fdc_sinks_group_destroy": (None, [_vp]),
    "fdc_sinks_group_work": (C.c_int, [_vp, _vp, C.c_int]),
    "fdc_sinks_group_pdu_count": (C.c_int, [_vp]),
    "fdc_sinks_group_pdus": (C.c_int, [_vp, C.POINTER(fdc_pdu), C.c_int]),
    "fdc_sinks_group_size": (C.c_int32, [_vp]),
    "fdc_sinks_group_member": (_vp, [_vp, C.c_int]),
    "fdc_sinks_group_member_info": (C.c_int, [_vp, C.c_int] + [C.POINTER(C.c_int32)] * 5),
    "fdc_sinks_spectrum": (_vp, [_vp]),
    "fdc_sinks_stream": (_vp, [_vp]),
    "fdc_sinks_spectrum_ahead": (_vp, [_vp]),
    "fdc_sinks_fill_stream": (_vp, [_vp]),
    "fdc_sinks_prepare": (C.c_int, [_vp, C.c_int, C.c_int]),
    "fdc_sinks_blocklen": (C.c_int32, [_vp]),
    "fdc_sinks_max_blocks": (C.c_int32, [_vp]),
    "fdc_sinks_work_device": (C.c_int, [_vp, C.c_int]),
    "fdc_sinks_submit_device": (C.c_int, [_vp, C.c_int]),
    "fdc_sinks_flush": (C.c_int, [_vp]),
    "fdc_sinks_engine": (C.c_int32, [_vp]),
    "fdc_set_log_callback": (None, [_vp, _vp]),
    "fdc_sinks_pdu_count": (C.c_int, [_vp]),
    "fdc_sinks_pdu": (C.c_int, [_vp, C.c_int, C.POINTER(fdc_pdu)]),
    "fdc_sinks_pdus": (C.c_int, [_vp, C.POINTER(fdc_pdu), C.c_int]),
    "fdc_sinks_pac_params": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int32)]),
    "fdc_sinks_segment_params": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int32)]),
    "fdc_overlap_save_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "fdc_overlap_save_work": (C.c_int, [_vp, _vp, C.c_int, _vp]),
    "fdc_overlap_save_destroy": (None, [_vp]),
    "fdc_vector_cut_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "fdc_vector_cut_work": (C.c_int, [_vp, _vp, C.c_int, _vp]),
    "fdc_vector_cut_destroy": (None, [_vp]),
    "fdc_phase_window_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int,
                                          C.POINTER(_vp)]),
    "fdc_phase_window_work": (C.c_int, [_vp, _vp, C.c_int, _vp]),
    "fdc_phase_window_destroy": (None, [_vp]),
    "fdc_window_table": (C.c_int, [C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, _vp]),
    "fdc_fft_vcc": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp]),
}

_lib = None


def lib():
    """Load libfdc_amd.so once.  Raises if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("gr-fdc_amd: %s is missing — build it with `make -C gr-fdc_amd/csrc` "
                              "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(h, name)      # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def check(status):
    if status < 0:
        raise FdcError(status, lib().fdc_last_error().decode())
    return status
