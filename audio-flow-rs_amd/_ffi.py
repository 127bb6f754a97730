"""ctypes binding of libaudioflow_dsp.so (the C ABI in include/audioflow_dsp.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).
There is no fallback: if the shared object is missing, loading raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# AF_LIB: another build of the SAME library (sanitizer / ablation builds of scripts/); the product is the in-tree file
LIB_PATH = os.environ.get("AF_LIB") or os.path.join(_HERE, "libaudioflow_dsp.so")
CSRC_DIR = os.path.join(_HERE, "csrc")

AF_OK = 0
AF_ERR_INVALID_ARG = 1
AF_ERR_BUFFER_TOO_SMALL = 2
AF_ERR_NO_DEVICE = 3
AF_ERR_HIP = 4
AF_ERR_OUT_OF_MEMORY = 5
AF_ERR_UNSUPPORTED = 6

AF_PCM_F32 = 0
AF_PCM_I16 = 1


class FeaturesConfig(C.Structure):
    _fields_ = [("sample_rate", C.c_uint32), ("channels", C.c_uint32), ("n_fft", C.c_uint32),
                ("hop", C.c_uint32), ("n_mels", C.c_uint32), ("f_min", C.c_float),
                ("f_max", C.c_float), ("sample_format", C.c_uint32), ("device", C.c_int32),
                ("output", C.c_uint32), ("log_mel", C.c_uint32), ("mel_scale", C.c_uint32),
                ("mel_norm", C.c_uint32), ("channel_mode", C.c_uint32)]


class StreamConfig(C.Structure):
    _fields_ = [("features", FeaturesConfig), ("step_frames", C.c_size_t)]


class BatchDesc(C.Structure):
    _fields_ = [("cfg", FeaturesConfig), ("n_clips", C.c_size_t), ("clip_offsets", C.c_void_p)]


class StftConfig(C.Structure):
    _fields_ = [("channels", C.c_uint32), ("n_fft", C.c_uint32), ("hop", C.c_uint32),
                ("sample_format", C.c_uint32), ("device", C.c_int32)]


class MelConfig(C.Structure):
    _fields_ = [("sample_rate", C.c_uint32), ("n_fft", C.c_uint32), ("n_mels", C.c_uint32),
                ("f_min", C.c_float), ("f_max", C.c_float), ("mel_scale", C.c_uint32),
                ("mel_norm", C.c_uint32), ("log_mel", C.c_uint32), ("device", C.c_int32)]


AF_OUT_MEL, AF_OUT_POWER = 0, 1
AF_MEL_HTK, AF_MEL_SLANEY = 0, 1
AF_MEL_NORM_NONE, AF_MEL_NORM_SLANEY = 0, 1
AF_CH_DOWNMIX, AF_CH_SEPARATE = 0, 1


class ChainConfig(C.Structure):
    _fields_ = [("input_rate", C.c_uint32), ("channels", C.c_uint32), ("output_rate", C.c_uint32),
                ("fir_taps", C.c_void_p), ("n_fir_taps", C.c_size_t), ("hop", C.c_uint32),
                ("n_mels", C.c_uint32), ("f_min", C.c_float), ("f_max", C.c_float),
                ("device", C.c_int32), ("sample_format", C.c_uint32)]


class WireConfig(C.Structure):
    _fields_ = [("input_rate", C.c_uint32), ("output_rate", C.c_uint32), ("channels", C.c_uint32),
                ("sample_format", C.c_uint32), ("frame_samples", C.c_uint32), ("device", C.c_int32)]


class VadConfig(C.Structure):
    _fields_ = [("threshold_db", C.c_float), ("smoothing_factor", C.c_float),
                ("silence_timeout_frames", C.c_size_t), ("min_speech_frames", C.c_size_t)]


_vp = C.c_void_p
_sz = C.c_size_t
_szp = C.POINTER(C.c_size_t)
_u32 = C.c_uint32
_st = C.c_int32

# name -> (restype, argtypes); mirrors include/audioflow_dsp.h one to one
SIGNATURES = {
    "af_last_error": (C.c_char_p, []),
    "af_version": (C.c_char_p, []),
    "af_device_count": (C.c_int, []),
    "af_debug_set": (C.c_int, [C.c_char_p, C.c_int]),
    # to_mono
    "af_to_mono_out_len": (_sz, [_sz, C.c_uint16]),
    "af_to_mono": (_st, [_vp, _sz, C.c_uint16, _vp, _sz, _szp]),
    "af_to_mono_device": (_st, [_vp, _sz, C.c_uint16, _vp, _vp]),
    # quantiser / base64
    "af_quantize_pcm16": (_st, [_vp, _sz, _vp]),
    "af_quantize_pcm16_device": (_st, [_vp, _sz, _vp, _vp]),
    "af_base64_len": (_sz, [_sz]),
    "af_pcm16_base64": (_st, [_vp, _sz, _vp, _sz, _szp]),
    "af_pcm16_base64_device": (_st, [_vp, _sz, _vp, _vp]),
    # VAD
    "af_vad_energy": (_st, [_vp, _sz, _sz, _vp]),
    "af_vad_energy_device": (_st, [_vp, _sz, _sz, _vp, _vp]),
    "af_level_meter": (_st, [_vp, _sz, _sz, _vp, _vp, _vp]),
    "af_level_meter_device": (_st, [_vp, _sz, _sz, _vp, _vp, _vp, _vp]),
    "af_vad_default_config": (None, [C.POINTER(VadConfig)]),
    "af_vad_create": (_st, [C.POINTER(VadConfig), C.POINTER(_vp)]),
    "af_vad_destroy": (None, [_vp]),
    "af_vad_detect": (_st, [_vp, _vp, _sz, C.POINTER(C.c_int)]),
    "af_vad_detect_frames": (_st, [_vp, _vp, _sz, _sz, _vp]),
    "af_vad_reset": (None, [_vp]),
    "af_resample_release_staging": (None, []),
    "af_resample_cubic_batch": (_st, [_u32, _u32, _vp, _sz, _sz, _vp, _sz]),
    "af_resample_cubic_batch_device": (_st, [_u32, _u32, _vp, _sz, _sz, _vp, _vp]),
    "af_vad_detect_batch_device": (_st, [C.POINTER(VadConfig), _vp, _sz, _sz, _sz, _vp, _vp, _sz, _vp]),
    "af_vad_detect_batch": (_st, [C.POINTER(VadConfig), _vp, _sz, _sz, _sz, _vp]),
    "af_vad_state": (C.c_int, [_vp]),
    "af_vad_energy_db": (C.c_float, [_vp]),
    "af_vad_speech_frame_count": (_sz, [_vp]),
    # AudioResampler
    "af_resampler_create": (_st, [_u32, _u32, C.POINTER(_vp)]),
    "af_resampler_create_48k_to_16k": (_st, [C.POINTER(_vp)]),
    "af_resampler_destroy": (None, [_vp]),
    "af_resampler_input_rate": (_u32, [_vp]),
    "af_resampler_output_rate": (_u32, [_vp]),
    "af_resampler_needs_resampling": (C.c_int, [_vp]),
    "af_resampler_out_len": (_sz, [_vp, _sz]),
    "af_resampler_process": (_st, [_vp, _vp, _sz, _vp, _sz, _szp]),
    "af_resampler_flush": (_st, [_vp, _vp, _sz, _szp]),
    "af_resampler_reset": (None, [_vp]),
    "af_resample_out_len": (_sz, [_u32, _u32, _sz]),
    "af_resample_batch": (_st, [_u32, _u32, _vp, _sz, _sz, _vp, _sz]),
    "af_resample_batch_device": (_st, [_u32, _u32, _vp, _sz, _sz, _vp, _vp]),
    "af_resample_batch_stereo_device": (_st, [_u32, _u32, _vp, _sz, _sz, _vp, _vp]),
    "af_resample_has_stereo_path": (C.c_int, [_u32, _u32]),
    "af_resample_prepare": (_st, [_u32, _u32]),
    "af_resample_plan_describe": (C.c_int, [_u32, _u32, _u32, _vp]),
    "af_chain_fused_plan_describe": (C.c_int, [_u32, _u32, _u32, _sz, _vp]),
    "af_chain_n_launches": (C.c_int, [_vp]),
    "af_resample_taps": (_st, [_u32, _u32, _vp, _sz, C.POINTER(_u32), C.POINTER(_u32),
                               C.POINTER(_u32)]),
    # BatchResampler
    "af_batch_resampler_create": (_st, [_u32, _u32, C.POINTER(_vp)]),
    "af_batch_resampler_destroy": (None, [_vp]),
    "af_batch_resampler_out_len": (_sz, [_vp, _sz]),
    "af_batch_resampler_buffered": (_sz, [_vp]),
    "af_batch_resampler_process": (_st, [_vp, _vp, _sz, _vp, _sz, _szp]),
    "af_batch_resampler_flush": (_st, [_vp, _vp, _sz, _szp]),
    # FIR
    "af_fir_create": (_st, [_vp, _sz, C.POINTER(_vp)]),
    "af_fir_destroy": (None, [_vp]),
    "af_fir_process": (_st, [_vp, _vp, _sz, _vp]),
    "af_fir_reset": (None, [_vp]),
    "af_fir_batch": (_st, [_vp, _sz, _vp, _sz, _sz, _vp]),
    "af_fir_apply_batch_device": (_st, [_vp, _vp, _sz, _sz, _vp, _vp]),
    "af_fir_batch_device": (_st, [_vp, _sz, _vp, _sz, _sz, _vp, _vp]),
    # FeaturePipeline
    "af_features_default_config": (None, [C.POINTER(FeaturesConfig)]),
    "af_features_create": (_st, [C.POINTER(FeaturesConfig), C.POINTER(_vp)]),
    "af_features_destroy": (None, [_vp]),
    "af_features_n_frames": (_sz, [_vp, _sz]),
    "af_features_row_len": (_sz, [_vp]),
    "af_features_set_batch": (_st, [_vp, _vp, _sz, _szp]),
    "af_features_set_batch_uniform": (_st, [_vp, _sz, _sz, _szp]),
    "af_features_set_batch_spans": (_st, [_vp, _vp, _vp, _sz, _szp]),
    "af_features_run_device": (_st, [_vp, _vp, _vp, _vp]),
    "af_features_run": (_st, [_vp, _vp, _sz, _vp, _sz]),
    "af_pipeline_run": (_st, [C.POINTER(BatchDesc), _vp, _sz, _vp, _sz, _szp]),
    "af_features_mel_weights": (_st, [_vp, _vp, _sz]),
    "af_features_kernel_name": (C.c_char_p, []),
    "af_features_plan_describe": (C.c_int, [_vp, _vp]),
    "af_features_kernel_describe": (C.c_int, [_vp, _vp]),
    "af_features_debug_stamps": (_st, [_vp, C.c_int, _vp]),
    # full chain
    "af_stft_default_config": (None, [C.POINTER(StftConfig)]),
    "af_stft_create": (_st, [C.POINTER(StftConfig), C.POINTER(_vp)]),
    "af_stft_destroy": (None, [_vp]),
    "af_stft_n_frames": (_sz, [_vp, _sz]),
    "af_stft_n_bins": (_sz, [_vp]),
    "af_stft_set_batch_uniform": (_st, [_vp, _sz, _sz, _szp]),
    "af_stft_set_batch": (_st, [_vp, _vp, _sz, _szp]),
    "af_stft_power_device": (_st, [_vp, _vp, _vp, _vp]),
    "af_stft_power": (_st, [_vp, _vp, _sz, _vp, _sz]),
    "af_mel_default_config": (None, [C.POINTER(MelConfig)]),
    "af_mel_create": (_st, [C.POINTER(MelConfig), C.POINTER(_vp)]),
    "af_mel_destroy": (None, [_vp]),
    "af_mel_weights": (_st, [_vp, _vp, _sz]),
    "af_mel_apply_device": (_st, [_vp, _vp, _sz, _vp, _vp]),
    "af_mel_apply": (_st, [_vp, _vp, _sz, _vp, _sz]),
    "af_chain_default_config": (None, [C.POINTER(ChainConfig)]),
    "af_chain_create": (_st, [C.POINTER(ChainConfig), C.POINTER(_vp)]),
    "af_chain_destroy": (None, [_vp]),
    "af_chain_n_frames": (_sz, [_vp, _sz]),
    "af_chain_set_batch_uniform": (_st, [_vp, _sz, _sz, _szp]),
    "af_chain_run_device": (_st, [_vp, _vp, _vp, _vp]),
    "af_chain_run": (_st, [_vp, _vp, _sz, _vp, _sz]),
    "af_chain_set_batch": (_st, [_vp, _vp, _sz, _szp]),
    "af_chain_set_option": (_st, [_vp, C.c_char_p, C.c_int]),
    "af_features_set_option": (_st, [_vp, C.c_char_p, C.c_int]),
    "af_chain_stream_create": (_st, [C.POINTER(ChainConfig), _sz, C.POINTER(_vp)]),
    "af_chain_stream_destroy": (None, [_vp]),
    "af_chain_stream_step_frames": (_sz, [_vp]),
    "af_chain_stream_rows_per_step": (_sz, [_vp]),
    "af_chain_stream_row_len": (_sz, [_vp]),
    "af_chain_stream_total_rows": (_sz, [_vp, C.c_uint64]),
    "af_chain_stream_flush_rows": (_sz, [_vp]),
    "af_chain_stream_max_rows": (_sz, [_vp, _sz]),
    "af_chain_stream_push": (_st, [_vp, _vp, _sz, _vp, _sz, _szp]),
    "af_chain_stream_flush": (_st, [_vp, _vp, _sz, _szp]),
    "af_chain_stream_reset": (_st, [_vp]),
    "af_chain_stream_stats": (_st, [_vp, _vp]),
    "af_chain_stream_first_step_frames": (_sz, [_vp]),
    "af_chain_stream_history_frames": (_sz, [_vp]),
    "af_chain_stream_seek": (_st, [_vp, C.c_uint64, C.POINTER(C.c_uint64)]),
    # WireEncoder: capture -> wire path in one launch (capture.rs:30-42, resampler.rs:60-93, websocket.rs:244-254)
    "af_wire_default_config": (None, [C.POINTER(WireConfig)]),
    "af_wire_create": (_st, [C.POINTER(WireConfig), C.POINTER(_vp)]),
    "af_wire_destroy": (None, [_vp]),
    "af_wire_out_samples": (_sz, [_vp, _sz]),
    "af_wire_text_len": (_sz, [_vp, _sz]),
    "af_wire_message_len": (_sz, [_vp]),
    "af_wire_n_messages": (_sz, [_vp, _sz]),
    "af_wire_set_batch_uniform": (_st, [_vp, _sz, _sz, _szp, _szp]),
    "af_wire_set_batch": (_st, [_vp, _vp, _sz, _szp, _szp]),
    "af_wire_text_offsets": (_st, [_vp, _vp, _sz]),
    "af_wire_total_messages": (_sz, [_vp]),
    "af_wire_run_device": (_st, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "af_wire_run": (_st, [_vp, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _vp, _sz]),
    "af_wire_set_option": (_st, [_vp, C.c_char_p, C.c_int]),
    "af_wire_n_launches": (C.c_int, [_vp]),
    # RingBuffer (capture.rs:84-161)
    "af_ring_create": (_st, [_sz, C.POINTER(_vp)]),
    "af_ring_destroy": (None, [_vp]),
    "af_ring_capacity": (_sz, [_vp]),
    "af_ring_write": (_sz, [_vp, _vp, _sz]),
    "af_ring_read": (_st, [_vp, _sz, _vp, _sz, _szp, C.POINTER(C.c_int)]),
    "af_ring_available": (_sz, [_vp]),
    "af_ring_clear": (None, [_vp]),
    # FeatureStream (BASELINE config 5)
    "af_stream_default_config": (None, [C.POINTER(StreamConfig)]),
    "af_stream_create": (_st, [C.POINTER(StreamConfig), C.POINTER(_vp)]),
    "af_stream_destroy": (None, [_vp]),
    "af_stream_rows_per_step": (_sz, [_vp]),
    "af_stream_row_len": (_sz, [_vp]),
    "af_stream_total_rows": (_sz, [_vp, C.c_uint64]),
    "af_stream_max_rows": (_sz, [_vp, _sz]),
    "af_stream_push": (_st, [_vp, _vp, _sz, _vp, _sz, _szp]),
    "af_stream_flush": (_st, [_vp, _vp, _sz, _szp]),
    "af_stream_reset": (_st, [_vp]),
    "af_stream_stats": (_st, [_vp, _vp]),
}

_lib = None


def load(path: str | None = None) -> C.CDLL:
    """Load libaudioflow_dsp.so and attach the declared signatures.

    Importing torch first (when it is going to be used for device memory) makes
    the dynamic loader reuse torch's already-loaded libamdhip64.so.7, so both
    sides share one HIP runtime and device pointers can cross the boundary.
    """
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise OSError(
            f"{p} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(p, mode=C.RTLD_GLOBAL)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError => header/library mismatch
        fn.restype = restype
        fn.argtypes = argtypes
    if path is None:
        _lib = lib
    return lib
