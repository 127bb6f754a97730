"""The C-ABI library loads without a GPU, exports every symbol that
include/audioflow_dsp.h declares, and fails loudly (no CPU fallback) when a
compute entry point is called without a device.  Pure-bookkeeping entry points
are checked for exact values."""
import ctypes as C
import os
import re

import numpy as np

from util import bits_equal
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu(af):
    return af.device_count() > 0


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "audioflow_dsp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(af_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(af):
    names = declared_symbols()
    assert len(names) >= 50
    assert sorted(af._ffi.SIGNATURES) == names


def test_every_declared_symbol_is_exported(af):
    lib = C.CDLL(af._ffi.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in include/audioflow_dsp.h but not exported"


def test_version_and_error_string(af):
    lib = af.library()
    assert b"gfx950" in lib.af_version()
    assert isinstance(lib.af_last_error(), bytes)


def test_length_bookkeeping_is_exact(af):
    lib = af.library()
    # to_mono: ceil(n/channels) (capture.rs:35 `.chunks`)
    assert lib.af_to_mono_out_len(4, 2) == 2
    assert lib.af_to_mono_out_len(5, 2) == 3
    assert lib.af_to_mono_out_len(7, 1) == 7
    assert lib.af_to_mono_out_len(7, 0) == 0
    # padded base64 (websocket.rs:254)
    for n, want in [(0, 0), (1, 4), (2, 4), (3, 4), (4, 8), (640, 856)]:
        assert lib.af_base64_len(n) == want
    # ceil(n*L/M) (docs/OPS.md)
    assert lib.af_resample_out_len(44100, 16000, 441000) == 160000
    assert lib.af_resample_out_len(48000, 16000, 128) == 43
    assert lib.af_resample_out_len(48000, 16000, 1440000) == 480000
    assert lib.af_resample_out_len(16000, 16000, 777) == 777
    assert lib.af_resample_out_len(48000, 16000, 0) == 0


def test_resampler_taps_match_oracle_design(af, po):
    # the device's prototype filter is the oracle's design rounded to f32 (host-only call)
    lib = af.library()
    L, M, T = C.c_uint32(), C.c_uint32(), C.c_uint32()
    assert lib.af_resample_taps(44100, 16000, None, 0, C.byref(L), C.byref(M), C.byref(T)) == 0
    assert (L.value, M.value, T.value) == (160, 441, 64)
    taps = np.empty(L.value * T.value, np.float32)
    assert lib.af_resample_taps(44100, 16000, taps.ctypes.data, taps.size, None, None, None) == 0
    want = po.resample_design(160, 441, 64).astype(np.float32)
    assert bits_equal(taps, want)
    small = np.empty(10, np.float32)
    assert lib.af_resample_taps(44100, 16000, small.ctypes.data, small.size, None, None, None) \
        == af._ffi.AF_ERR_BUFFER_TOO_SMALL


def test_invalid_arguments_are_reported(af):
    lib = af.library()
    h = C.c_void_p()
    assert lib.af_resampler_create(0, 16000, C.byref(h)) == af._ffi.AF_ERR_INVALID_ARG
    assert b"positive" in lib.af_last_error()
    assert lib.af_fir_create(None, 0, C.byref(h)) == af._ffi.AF_ERR_INVALID_ARG
    with pytest.raises(af.AudioError) as e:
        af.AudioResampler.new(0, 16000)
    assert e.value.kind == "ResamplingFailed"                 # resampler.rs:49 error mapping
    assert str(e.value).startswith("Resampling failed: ")     # error.rs:109


def test_equal_rates_passthrough_needs_no_device(af):
    # resampler.rs:185-190 test_no_resample_needed: bit-identical copy (no arithmetic at all)
    r = af.AudioResampler.new(16000, 16000)
    x = np.array([0.1, 0.2, 0.3, 0.4], np.float32)
    assert bits_equal(r.process(x), x)
    assert not r.needs_resampling()                            # resampler.rs:201-204
    assert r.input_rate() == 16000 and r.output_rate() == 16000


def test_no_cpu_fallback_without_device(af):
    if _has_gpu(af):
        pytest.skip("a GPU is present; the no-device path cannot be observed here")
    lib = af.library()
    with pytest.raises(af.AudioError) as e:
        af.FeaturePipeline()
    assert e.value.status == af._ffi.AF_ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)
    with pytest.raises(af.AudioError):
        af.AudioResampler.new(48000, 16000)                    # unequal rates need the device
    with pytest.raises(af.AudioError):
        af.to_mono([0.5, 0.25, -0.5, -0.25], 2)
    with pytest.raises(af.AudioError):
        af.quantize_pcm16([0.5])
    with pytest.raises(af.AudioError):
        af.VoiceActivityDetector().detect(np.zeros(480, np.float32))
    assert lib.af_device_count() == 0
