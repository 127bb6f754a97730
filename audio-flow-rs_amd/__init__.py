"""audio_flow_rs_amd -- host-side mirror of audio-flow's audio module over the
MI355X C ABI (``libaudioflow_dsp.so``, ``include/audioflow_dsp.h``).

The directory is called ``audio-flow-rs_amd`` (not importable as written); the
repo-root shim ``audio_flow_rs_amd.py`` loads it under this module name.

Names, argument meaning and error behaviour follow the reference
(forfd8960/audio-flow-rs, ``src-tauri/src/modules/audio/*.rs``):

=========================  ===================================================
``AudioFrame``             capture.rs:13-43 (``to_mono`` :30)
``RingBuffer``             capture.rs:84-161
``AudioResampler``         resampler.rs:23-109
``BatchResampler``         resampler.rs:114-167
``VadConfig/VadState/``    vad.rs:21-205
``VoiceActivityDetector``
``MessageBuilder``         network/websocket.rs:333-363 (audio quantiser)
``WireEncoder``            capture.rs:30-42 + resampler.rs:60-93 + websocket.rs:244-254 in one launch
``AudioError``             error.rs:96-111
``FirFilter`` / ``Stft`` / build-defined siblings in the same style
``MelFilterbank`` /        (docs/OPS.md)
``FeaturePipeline``
=========================  ===================================================

All arithmetic runs in hand-written HIP kernels on gfx950; there is no CPU
fallback and nothing here imports the test oracle.
"""
from . import _ffi
from .errors import AudioError
from .audio import (AudioFrame, AudioResampler, BatchResampler, ChainPipeline, ChainStream, FeaturePipeline, FeatureStream, FirFilter,
                    MelSpectrogram, MessageBuilder, RingBuffer, VadConfig, VadLevel, VadState,
                    VoiceActivityDetector, device_count, library, quantize_pcm16, to_mono,
                    vad_energy, resample_batch, fir_batch, AudioLevel, level_meter, Stft, MelFilterbank, resample_stereo_batch,
                    resample_cubic_batch, WireEncoder)

__all__ = [
    "AudioError", "AudioFrame", "AudioResampler", "BatchResampler", "ChainPipeline", "ChainStream", "FeaturePipeline", "FeatureStream",
    "FirFilter",
    "MelSpectrogram", "MessageBuilder", "RingBuffer", "VadConfig", "VadLevel", "VadState",
    "VoiceActivityDetector", "device_count", "library", "quantize_pcm16", "to_mono", "vad_energy",
    "resample_batch", "fir_batch", "AudioLevel", "level_meter", "Stft", "MelFilterbank", "resample_stereo_batch", "resample_cubic_batch", "WireEncoder",
]
