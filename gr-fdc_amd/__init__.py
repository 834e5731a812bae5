"""gr-fdc_amd — MI355X-native frequency-domain channelizer (the gr-FDC overlap-save hot path).

Python host side: a mirror of the reference's Python face (python/FrequencyDomainChannelizer.py and the
SWIG block factories of swig/FDC_swig.i) over the C-ABI in include/fdc_amd.h.  All signal processing
runs in hand-written gfx950 HIP kernels (gr-fdc_amd/csrc); nothing here computes on the CPU.
"""
from ._lib import FdcError, lib, LIB_PATH                                   # noqa: F401
from .blocks import overlap_save, vector_cut_vxx, phase_shifting_windowing_vcc, fft_vcc, window_table  # noqa: F401
from .channelizer import (FrequencyDomainChannelizer, Pipeline, PipelineGroup, plan_preview, FREQMODE, VERBOSEMODE, WINDOWTYPES,   # noqa: F401
                          nextpow2, get_opt_channelparams, freq_converters, register_host, unregister_host, defaults)
from ._lib import (FDC_PIPE_FORCE_GENERIC, FDC_PIPE_NO_POLY, FDC_PIPE_NO_BLOCK, FDC_PIPE_PLAIN_STORES, FDC_PIPE_NT_LOADS,   # noqa: F401
                   FDC_PIPE_FULL_SPECTRUM, FDC_PIPE_WIDE_UNIFORM, FDC_PIPE_NO_FUSED,
                   FDC_SINKS_HOST_DECISIONS, FDC_SINKS_DEVICE_PAYLOAD, FDC_SINKS_LOOKAHEAD)
from .sharding import span_for_rank, ring_bounds, ring_for_span                       # noqa: F401
from .sinks import Sinks, SinksGroup, PowerActivationChannel, activity_detection_channelizer_vcm, SegmentDetection      # noqa: F401
