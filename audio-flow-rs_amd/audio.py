"""Host-side mirror of the reference's audio module over the C ABI.

Every class keeps the reference's method names and semantics and cites the
lines it mirrors (paths relative to forfd8960/audio-flow-rs ``src-tauri/src/``).
numpy arrays stand in for ``Vec<f32>``; failures raise :class:`AudioError`.
"""
from __future__ import annotations

import ctypes as C
import enum
import threading
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import _ffi
from .errors import AudioError


def library() -> C.CDLL:
    """The loaded C-ABI library (raises OSError if it has not been built)."""
    return _ffi.load()


def _check(status: int) -> None:
    if status != _ffi.AF_OK:
        msg = library().af_last_error().decode("utf-8", "replace")
        raise AudioError.resampling_failed(msg, status)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32).reshape(-1)


def _ptr(a: np.ndarray) -> int:
    return a.ctypes.data


def device_count() -> int:
    return int(library().af_device_count())


# --------------------------------------------------------------------------
# free functions (device kernels behind one call)
def to_mono(samples, channels: int) -> np.ndarray:
    """capture.rs:30-42 on the device."""
    x = _f32(samples)
    lib = library()
    if channels <= 0:
        raise AudioError.resampling_failed("to_mono: channels must be > 0", _ffi.AF_ERR_INVALID_ARG)
    out = np.empty(lib.af_to_mono_out_len(x.size, channels), np.float32)
    n = C.c_size_t()
    _check(lib.af_to_mono(_ptr(x), x.size, channels, _ptr(out), out.size, C.byref(n)))
    return out[: n.value]


def quantize_pcm16(samples) -> np.ndarray:
    """network/websocket.rs:246-251: f32 -> i16 (clamp, *32767, truncate)."""
    x = _f32(samples)
    out = np.empty(x.size, np.int16)
    _check(library().af_quantize_pcm16(_ptr(x), x.size, _ptr(out)))
    return out


def vad_energy(frames, frame_len: int) -> np.ndarray:
    """vad.rs:157-168 for consecutive frames of ``frame_len`` samples."""
    x = _f32(frames)
    n_frames = x.size // frame_len if frame_len else 0
    out = np.empty(n_frames, np.float32)
    _check(library().af_vad_energy(_ptr(x), n_frames, frame_len, _ptr(out)))
    return out


@dataclass
class AudioLevel:
    """specs/0002-design.md:727-732"""
    rms: float
    peak: float
    dbfs: float


def level_meter(samples, window_len: int) -> List[AudioLevel]:
    """rms / peak / dBFS of consecutive windows (the ``audio_level`` event payload,
    events/mod.rs:73-75; spec'd every 50 ms = 800 samples at 16 kHz)."""
    x = _f32(samples)
    n = x.size // window_len if window_len else 0
    rms, peak, dbfs = (np.empty(max(n, 1), np.float32) for _ in range(3))
    _check(library().af_level_meter(_ptr(x), n, window_len, _ptr(rms), _ptr(peak), _ptr(dbfs)))
    return [AudioLevel(float(rms[i]), float(peak[i]), float(dbfs[i])) for i in range(n)]


def resample_batch(clips, input_rate: int, output_rate: int) -> np.ndarray:
    """One-shot polyphase resample of equal-length mono clips ``[n_clips, n_in]``."""
    x = np.ascontiguousarray(clips, np.float32)
    if x.ndim == 1:
        x = x[None, :]
    lib = library()
    n_out = lib.af_resample_out_len(input_rate, output_rate, x.shape[1])
    out = np.empty((x.shape[0], n_out), np.float32)
    _check(lib.af_resample_batch(input_rate, output_rate, _ptr(x), x.shape[0], x.shape[1],
                                 _ptr(out), out.size))
    return out


def resample_cubic_batch(clips, input_rate: int, output_rate: int) -> np.ndarray:
    """Optional cubic mode [B, unpinned]: what resampler.rs:43-49 configures (rubato FastFixedIn, Cubic) as defined
    in docs/OPS.md "Cubic interpolation"; equal-length mono clips ``[n_clips, n_in]``."""
    x = np.ascontiguousarray(clips, np.float32)
    if x.ndim == 1:
        x = x[None, :]
    lib = library()
    n_out = lib.af_resample_out_len(input_rate, output_rate, x.shape[1])
    out = np.empty((x.shape[0], n_out), np.float32)
    _check(lib.af_resample_cubic_batch(input_rate, output_rate, _ptr(x), x.shape[0], x.shape[1], _ptr(out), out.size))
    return out


def resample_stereo_batch(clips, input_rate: int, output_rate: int) -> np.ndarray:
    """``to_mono`` fused into the resampler (af_resample_batch_stereo_device): ``clips`` is
    ``[n_clips, n_frames * 2]`` interleaved stereo; returns ``[n_clips, ceil(n_frames L / M)]``."""
    import ctypes as _C
    x = np.ascontiguousarray(clips, np.float32)
    if x.ndim == 1:
        x = x[None, :]
    n_clips, n_frames = x.shape[0], x.shape[1] // 2
    lib = library()
    n_out = int(lib.af_resample_out_len(input_rate, output_rate, n_frames))
    import torch
    dev = torch.device("cuda", torch.cuda.current_device())
    xin = torch.from_numpy(x).to(dev)
    out = torch.empty(n_clips, n_out, device=dev, dtype=torch.float32)
    _check(lib.af_resample_batch_stereo_device(input_rate, output_rate, _C.c_void_p(xin.data_ptr()),
                                               n_clips, n_frames, _C.c_void_p(out.data_ptr()), None))
    torch.cuda.synchronize()
    return out.cpu().numpy()


def fir_batch(clips, taps) -> np.ndarray:
    x = np.ascontiguousarray(clips, np.float32)
    if x.ndim == 1:
        x = x[None, :]
    h = _f32(taps)
    out = np.empty_like(x)
    _check(library().af_fir_batch(_ptr(h), h.size, _ptr(x), x.shape[0], x.shape[1], _ptr(out)))
    return out


# --------------------------------------------------------------------------
@dataclass
class AudioFrame:
    """capture.rs:13-28: interleaved f32 PCM in [-1, 1], frame-major."""
    samples: np.ndarray
    sample_rate: int
    channels: int
    timestamp_ns: int = 0

    def __post_init__(self):
        self.samples = _f32(self.samples)

    @classmethod
    def new(cls, samples, sample_rate: int, channels: int, timestamp_ns: int = 0) -> "AudioFrame":
        return cls(samples, sample_rate, channels, timestamp_ns)          # capture.rs:25

    def to_mono(self) -> "AudioFrame":
        """capture.rs:30-42."""
        if self.channels == 1:
            return AudioFrame(self.samples.copy(), self.sample_rate, 1, self.timestamp_ns)
        return AudioFrame(to_mono(self.samples, self.channels), self.sample_rate, 1,
                          self.timestamp_ns)

    def clone(self) -> "AudioFrame":
        return AudioFrame(self.samples.copy(), self.sample_rate, self.channels, self.timestamp_ns)


class RingBuffer:
    """capture.rs:84-161: mutex-guarded f32 ring with ``capacity - 1`` usable slots, behind the C ABI
    (``af_ring_*``; pure host bookkeeping, it feeds the device path)."""

    def __init__(self, capacity_samples: int):
        self._lib = library()
        h = C.c_void_p()
        _check(self._lib.af_ring_create(int(capacity_samples), C.byref(h)))    # capture.rs:93-100
        self._h = h
        self.capacity = int(capacity_samples)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.af_ring_destroy(h)

    def write(self, data) -> int:
        """capture.rs:103-122: samples actually stored"""
        x = _f32(data)
        return int(self._lib.af_ring_write(self._h, _ptr(x), x.size))

    def read(self, size: int) -> Optional[np.ndarray]:
        """capture.rs:125-145 (``None`` when empty)"""
        out = np.empty(min(int(size), self.capacity), np.float32)
        n, none = C.c_size_t(), C.c_int()
        _check(self._lib.af_ring_read(self._h, int(size), _ptr(out), out.size, C.byref(n), C.byref(none)))
        return None if none.value else out[:n.value].copy()

    def available(self) -> int:
        return int(self._lib.af_ring_available(self._h))                   # capture.rs:148-152

    def clear(self) -> None:
        self._lib.af_ring_clear(self._h)                                    # capture.rs:155-160


# --------------------------------------------------------------------------
class AudioResampler:
    """resampler.rs:23-109.  ``process`` is streaming, like the rubato-backed
    original: it returns the outputs whose input support has arrived."""

    def __init__(self, input_rate: int, output_rate: int):
        self._lib = library()
        h = C.c_void_p()
        _check(self._lib.af_resampler_create(int(input_rate), int(output_rate), C.byref(h)))
        self._h = h

    @classmethod
    def new(cls, input_rate: int, output_rate: int) -> "AudioResampler":
        return cls(input_rate, output_rate)                          # resampler.rs:32

    @classmethod
    def create_48k_to_16k(cls) -> "AudioResampler":
        return cls(48000, 16000)                                     # resampler.rs:60

    @classmethod
    def default(cls) -> "AudioResampler":
        return cls(48000, 16000)                                     # resampler.rs:169-178

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.af_resampler_destroy(h)

    def process(self, input) -> np.ndarray:
        """resampler.rs:71-93"""
        x = _f32(input)
        cap = self._lib.af_resampler_out_len(self._h, x.size)
        out = np.empty(max(cap, 1), np.float32)
        n = C.c_size_t()
        _check(self._lib.af_resampler_process(self._h, _ptr(x), x.size, _ptr(out), cap, C.byref(n)))
        return out[: n.value].copy()

    def flush(self) -> np.ndarray:
        """[B] zero-extend the stream: totals ceil(n_in * L / M) outputs."""
        cap = self._lib.af_resampler_out_len(self._h, 0)
        out = np.empty(max(cap, 1), np.float32)
        n = C.c_size_t()
        _check(self._lib.af_resampler_flush(self._h, _ptr(out), cap, C.byref(n)))
        return out[: n.value].copy()

    def reset(self) -> None:
        self._lib.af_resampler_reset(self._h)

    def input_rate(self) -> int:
        return int(self._lib.af_resampler_input_rate(self._h))       # resampler.rs:96

    def output_rate(self) -> int:
        return int(self._lib.af_resampler_output_rate(self._h))      # resampler.rs:101

    def needs_resampling(self) -> bool:
        return bool(self._lib.af_resampler_needs_resampling(self._h))  # resampler.rs:106


class BatchResampler:
    """resampler.rs:114-167: 128-frame chunker with carry and zero-padded flush."""

    def __init__(self, input_rate: int, output_rate: int):
        self._lib = library()
        h = C.c_void_p()
        _check(self._lib.af_batch_resampler_create(int(input_rate), int(output_rate), C.byref(h)))
        self._h = h

    @classmethod
    def new(cls, input_rate: int, output_rate: int) -> "BatchResampler":
        return cls(input_rate, output_rate)                          # resampler.rs:121

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.af_batch_resampler_destroy(h)

    def buffered(self) -> int:
        return int(self._lib.af_batch_resampler_buffered(self._h))

    def process(self, input) -> np.ndarray:
        """resampler.rs:132-147"""
        x = _f32(input)
        cap = self._lib.af_batch_resampler_out_len(self._h, x.size)
        out = np.empty(max(cap, 1), np.float32)
        n = C.c_size_t()
        _check(self._lib.af_batch_resampler_process(self._h, _ptr(x), x.size, _ptr(out), cap,
                                                    C.byref(n)))
        return out[: n.value].copy()

    def flush(self) -> np.ndarray:
        """resampler.rs:150-166"""
        cap = self._lib.af_batch_resampler_out_len(self._h, 0)
        out = np.empty(max(cap, 1), np.float32)
        n = C.c_size_t()
        _check(self._lib.af_batch_resampler_flush(self._h, _ptr(out), cap, C.byref(n)))
        return out[: n.value].copy()


# --------------------------------------------------------------------------
class VadLevel(enum.Enum):
    """vad.rs:8-17"""
    Aggressive = 0
    Balanced = 1
    Relaxed = 2


class VadState(enum.IntEnum):
    """vad.rs:46-54"""
    Silence = 0
    Speech = 1
    Ending = 2


@dataclass
class VadConfig:
    """vad.rs:21-43 (defaults :34-43)"""
    threshold_db: float = -50.0
    smoothing_factor: float = 0.3
    silence_timeout_frames: int = 15
    min_speech_frames: int = 3


class VoiceActivityDetector:
    """vad.rs:60-205.  Energy on the device, state machine in the library."""

    def __init__(self, config: Optional[VadConfig] = None):
        self._lib = library()
        cfg = config or VadConfig()
        c = _ffi.VadConfig(cfg.threshold_db, cfg.smoothing_factor, cfg.silence_timeout_frames,
                           cfg.min_speech_frames)
        h = C.c_void_p()
        _check(self._lib.af_vad_create(C.byref(c), C.byref(h)))
        self._h = h

    @classmethod
    def new(cls, config: VadConfig) -> "VoiceActivityDetector":
        return cls(config)                                            # vad.rs:80

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.af_vad_destroy(h)

    def detect(self, frame) -> VadState:
        """vad.rs:97-154"""
        x = _f32(frame)
        st = C.c_int()
        _check(self._lib.af_vad_detect(self._h, _ptr(x), x.size, C.byref(st)))
        return VadState(st.value)

    def detect_frames(self, samples, frame_len: int) -> List[VadState]:
        x = _f32(samples)
        n_frames = x.size // frame_len if frame_len else 0
        out = np.zeros(max(n_frames, 1), np.int32)
        _check(self._lib.af_vad_detect_frames(self._h, _ptr(x), n_frames, frame_len, _ptr(out)))
        return [VadState(int(v)) for v in out[:n_frames]]

    def reset(self) -> None:
        self._lib.af_vad_reset(self._h)                               # vad.rs:179

    @staticmethod
    def detect_batch(clips, frame_len: int, config: Optional[VadConfig] = None) -> np.ndarray:
        """A batch of independent detectors (one per clip, vad.rs:60-70), scanned on the device:
        clips [n_clips][n_frames * frame_len] f32 -> states [n_clips][n_frames] (VadState values, uint8).
        Every row equals VoiceActivityDetector(config).detect_frames(row, frame_len)."""
        x = np.ascontiguousarray(clips, dtype=np.float32)
        assert x.ndim == 2 and frame_len > 0 and x.shape[1] % frame_len == 0
        n_clips, n_frames = x.shape[0], x.shape[1] // frame_len
        cfg = config or VadConfig()
        c = _ffi.VadConfig(cfg.threshold_db, cfg.smoothing_factor, cfg.silence_timeout_frames,
                           cfg.min_speech_frames)
        out = np.zeros((n_clips, max(n_frames, 1)), np.uint8)
        _check(library().af_vad_detect_batch(C.byref(c), _ptr(x), n_clips, n_frames, frame_len, _ptr(out)))
        return out[:, :n_frames]

    @staticmethod
    def detect_batch_device(energy_ptr: int, n_clips: int, n_frames: int, states_ptr: int,
                            config: Optional[VadConfig] = None, lanes_ptr: int = 0, row_stride: int = 0,
                            state_stride: int = 0, stream: int = 0) -> None:
        """Device pointers: energies [n_clips][row_stride] f32 -> states [n_clips][state_stride] uint8;
        lanes_ptr: optional [n_clips] af_vad_lane records carried between calls (streaming)."""
        cfg = config or VadConfig()
        c = _ffi.VadConfig(cfg.threshold_db, cfg.smoothing_factor, cfg.silence_timeout_frames,
                           cfg.min_speech_frames)
        _check(library().af_vad_detect_batch_device(C.byref(c), energy_ptr, n_clips, n_frames,
                                                    row_stride or n_frames, lanes_ptr or None, states_ptr,
                                                    state_stride or n_frames, stream or None))

    def state(self) -> VadState:
        return VadState(self._lib.af_vad_state(self._h))              # vad.rs:187

    def energy_db(self) -> float:
        return float(self._lib.af_vad_energy_db(self._h))             # vad.rs:192

    def is_speaking(self) -> bool:
        return self.state() == VadState.Speech                        # vad.rs:197

    def speech_frame_count(self) -> int:
        return int(self._lib.af_vad_speech_frame_count(self._h))      # vad.rs:202


# --------------------------------------------------------------------------
class MessageBuilder:
    """network/websocket.rs:333-363 (audio part): quantise + base64 on the device."""

    @staticmethod
    def audio_base64(audio_data) -> str:
        x = _f32(audio_data)
        lib = library()
        cap = lib.af_base64_len(2 * x.size)
        buf = np.zeros(max(cap, 1), np.uint8)
        n = C.c_size_t()
        _check(lib.af_pcm16_base64(_ptr(x), x.size, _ptr(buf), cap, C.byref(n)))
        return buf[: n.value].tobytes().decode("ascii")

    @staticmethod
    def audio_message(audio_data) -> str:
        """websocket.rs:338-352 (serde_json::json! sorts keys alphabetically)."""
        b64 = MessageBuilder.audio_base64(audio_data)
        return '{"audio_base_64":"%s","message_type":"input_audio_chunk"}' % b64

    @staticmethod
    def configure_message(model_id: str, language_code: str) -> str:
        """websocket.rs:355-362"""
        import json
        return json.dumps({"encoding": "pcm_16000", "language_code": language_code,
                           "message_type": "configure", "model_id": model_id},
                          separators=(",", ":"))


# --------------------------------------------------------------------------
class FirFilter:
    """docs/OPS.md "FIR" [B]; streaming like AudioResampler.process."""

    def __init__(self, taps):
        self._lib = library()
        h = _f32(taps)
        p = C.c_void_p()
        _check(self._lib.af_fir_create(_ptr(h), h.size, C.byref(p)))
        self._h = p
        self.n_taps = h.size

    @classmethod
    def new(cls, taps) -> "FirFilter":
        return cls(taps)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.af_fir_destroy(h)

    def process(self, input) -> np.ndarray:
        x = _f32(input)
        out = np.empty(x.size, np.float32)
        _check(self._lib.af_fir_process(self._h, _ptr(x), x.size, _ptr(out)))
        return out

    def reset(self) -> None:
        self._lib.af_fir_reset(self._h)


@dataclass
class MelSpectrogram:
    """One clip's ``[n_frames, n_mels]`` mel power spectrogram."""
    data: np.ndarray
    sample_rate: int
    hop: int


class FeaturePipeline:
    """docs/OPS.md "STFT"/"Mel" [B]: to_mono -> Hann STFT -> |X|^2 -> mel, fused."""

    def __init__(self, sample_rate: int = 48000, channels: int = 2, n_fft: int = 1024,
                 hop: int = 256, n_mels: int = 80, f_min: float = 0.0, f_max: float = 0.0,
                 sample_format: int = _ffi.AF_PCM_F32, device: int = -1, *,
                 output: int = _ffi.AF_OUT_MEL, log_mel: bool = False,
                 mel_scale: int = _ffi.AF_MEL_HTK, mel_norm: int = _ffi.AF_MEL_NORM_NONE,
                 channel_mode: int = _ffi.AF_CH_DOWNMIX):
        self._lib = library()
        self.cfg = _ffi.FeaturesConfig(sample_rate, channels, n_fft, hop, n_mels, f_min, f_max,
                                       sample_format, device, output, int(log_mel), mel_scale,
                                       mel_norm, channel_mode)
        h = C.c_void_p()
        _check(self._lib.af_features_create(C.byref(self.cfg), C.byref(h)))
        self._h = h
        self.total_frames = 0

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.af_features_destroy(h)

    # -- geometry
    def n_frames(self, n_pcm_frames: int) -> int:
        return int(self._lib.af_features_n_frames(self._h, n_pcm_frames))

    def row_len(self) -> int:
        """floats per output row: n_mels, or n_fft/2+1 with ``output=AF_OUT_POWER``"""
        return int(self._lib.af_features_row_len(self._h))

    def mel_weights(self) -> np.ndarray:
        W = np.empty((self.cfg.n_mels, self.cfg.n_fft // 2 + 1), np.float32)
        _check(self._lib.af_features_mel_weights(self._h, _ptr(W), W.size))
        return W

    # -- batch description
    def set_batch(self, clip_frame_offsets: Sequence[int]) -> int:
        off = np.ascontiguousarray(clip_frame_offsets, np.int64)
        tot = C.c_size_t()
        _check(self._lib.af_features_set_batch(self._h, _ptr(off), off.size - 1, C.byref(tot)))
        self.total_frames = tot.value
        return self.total_frames

    def set_batch_uniform(self, n_clips: int, frames_per_clip: int) -> int:
        tot = C.c_size_t()
        _check(self._lib.af_features_set_batch_uniform(self._h, n_clips, frames_per_clip,
                                                       C.byref(tot)))
        self.total_frames = tot.value
        return self.total_frames

    def set_batch_spans(self, starts: Sequence[int], lengths: Sequence[int]) -> int:
        """Spans ``[start, start + length)`` of one PCM buffer; they may overlap (streaming)."""
        st = np.ascontiguousarray(starts, np.int64)
        ln = np.ascontiguousarray(lengths, np.int64)
        tot = C.c_size_t()
        _check(self._lib.af_features_set_batch_spans(self._h, _ptr(st), _ptr(ln), st.size,
                                                     C.byref(tot)))
        self.total_frames = tot.value
        return self.total_frames

    @staticmethod
    def stream_spans(n_total: int, chunk: int, n_fft: int = 1024, hop: int = 256):
        """Cut a stream of ``n_total`` frames into chunks of ``chunk`` frames (a multiple of
        16*hop) with a right halo of ``n_fft - hop``: frame f belongs to the chunk that holds
        sample f*hop, so the concatenated chunk outputs are the one-shot framing."""
        if chunk % (16 * hop) != 0:
            raise ValueError("chunk must be a multiple of 16*hop")
        starts, lengths = [], []
        pos = 0
        while pos < n_total:
            ln = min(chunk + n_fft - hop, n_total - pos)
            starts.append(pos)
            lengths.append(ln)
            pos += chunk
        return starts, lengths

    # -- execution
    def run_device(self, pcm_ptr: int, mel_ptr: int, stream: int = 0) -> None:
        """Enqueue on ``stream`` (a hipStream_t as int); device pointers as ints."""
        _check(self._lib.af_features_run_device(self._h, C.c_void_p(pcm_ptr), C.c_void_p(mel_ptr),
                                                C.c_void_p(stream)))

    def run(self, pcm: np.ndarray) -> np.ndarray:
        """Host buffers: ``pcm`` is the batch's interleaved PCM back to back."""
        dt = np.int16 if self.cfg.sample_format == _ffi.AF_PCM_I16 else np.float32
        x = np.ascontiguousarray(pcm, dt).reshape(-1)
        mel = np.empty((self.total_frames, self.row_len()), np.float32)
        _check(self._lib.af_features_run(self._h, _ptr(x), x.size, _ptr(mel), mel.size))
        return mel

    def process_batch(self, frames: Sequence[AudioFrame]) -> List[MelSpectrogram]:
        """``FeaturePipeline::process_batch(&[AudioFrame]) -> Vec<MelSpectrogram>``."""
        ch = self.cfg.channels
        for f in frames:
            if f.channels != ch or f.sample_rate != self.cfg.sample_rate:
                raise AudioError.resampling_failed(
                    f"frame is {f.sample_rate} Hz x{f.channels}, pipeline expects "
                    f"{self.cfg.sample_rate} Hz x{ch}", _ffi.AF_ERR_INVALID_ARG)
        offs = np.zeros(len(frames) + 1, np.int64)
        for i, f in enumerate(frames):
            offs[i + 1] = offs[i] + f.samples.size // ch
        self.set_batch(offs)
        pcm = (np.concatenate([f.samples[: (f.samples.size // ch) * ch] for f in frames])
               if frames else np.zeros(0, np.float32))
        mel = self.run(pcm)
        # AF_CH_SEPARATE: a clip's rows are its channel-0 spectrogram followed by its channel-1 one
        per_clip = 2 if self.cfg.channel_mode == _ffi.AF_CH_SEPARATE else 1
        out, row = [], 0
        for i in range(len(frames)):
            nf = per_clip * self.n_frames(int(offs[i + 1] - offs[i]))
            out.append(MelSpectrogram(mel[row:row + nf], self.cfg.sample_rate, self.cfg.hop))
            row += nf
        return out


class FeatureStream:
    """The fused path over an unbounded stream in fixed steps (BASELINE config 5, ``af_stream_*``):
    ``push`` returns the rows of every step it completed (one hipGraph replay each), ``flush`` the rows
    of the last partial step; all rows in order == one-shot framing of the whole stream, bit for bit."""

    def __init__(self, step_frames: int = 1 << 20, **features):
        self._lib = library()
        cfg = _ffi.StreamConfig()
        self._lib.af_stream_default_config(C.byref(cfg))
        for k, v in features.items():
            setattr(cfg.features, k, v)
        cfg.step_frames = int(step_frames)
        self.cfg = cfg
        h = C.c_void_p()
        _check(self._lib.af_stream_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self.row_len = int(self._lib.af_stream_row_len(h))
        self.rows_per_step = int(self._lib.af_stream_rows_per_step(h))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.af_stream_destroy(h)

    def _dtype(self):
        return np.int16 if self.cfg.features.sample_format == _ffi.AF_PCM_I16 else np.float32

    def push(self, pcm) -> np.ndarray:
        x = np.ascontiguousarray(pcm, self._dtype()).reshape(-1)
        n_frames = x.size // self.cfg.features.channels
        rows = np.empty((int(self._lib.af_stream_max_rows(self._h, n_frames)), self.row_len), np.float32)
        n = C.c_size_t()
        _check(self._lib.af_stream_push(self._h, _ptr(x), n_frames, _ptr(rows), rows.size, C.byref(n)))
        return rows[:n.value]

    def flush(self) -> np.ndarray:
        rows = np.empty((self.rows_per_step, self.row_len), np.float32)
        n = C.c_size_t()
        _check(self._lib.af_stream_flush(self._h, _ptr(rows), rows.size, C.byref(n)))
        return rows[:n.value].copy()

    def reset(self) -> None:
        _check(self._lib.af_stream_reset(self._h))

    def total_rows(self, n_frames: int) -> int:
        return int(self._lib.af_stream_total_rows(self._h, int(n_frames)))

    def stats(self) -> dict:
        v = (C.c_double * 4)()
        _check(self._lib.af_stream_stats(self._h, v))
        return {"device_ms": v[0], "wall_ms": v[1], "steps": int(v[2]), "rows": int(v[3])}


class Stft:
    """``Stft::new(n_fft, hop)`` [B]: power spectrogram rows ``[frame][n_fft/2+1]``
    (docs/OPS.md "STFT"; BASELINE config 1 is ``Stft().power(x)`` on 160 000 samples)."""

    def __init__(self, n_fft: int = 1024, hop: int = 256, channels: int = 1,
                 sample_format: int = _ffi.AF_PCM_F32, device: int = -1):
        self._lib = library()
        self.cfg = _ffi.StftConfig(channels, n_fft, hop, sample_format, device)
        h = C.c_void_p()
        _check(self._lib.af_stft_create(C.byref(self.cfg), C.byref(h)))
        self._h = h
        self.total_frames = 0

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.af_stft_destroy(h)

    def n_frames(self, n_pcm_frames: int) -> int:
        return int(self._lib.af_stft_n_frames(self._h, n_pcm_frames))

    def n_bins(self) -> int:
        return int(self._lib.af_stft_n_bins(self._h))

    def set_batch_uniform(self, n_clips: int, frames_per_clip: int) -> int:
        tot = C.c_size_t()
        _check(self._lib.af_stft_set_batch_uniform(self._h, n_clips, frames_per_clip, C.byref(tot)))
        self.total_frames = tot.value
        return self.total_frames

    def set_batch(self, clip_frame_offsets: Sequence[int]) -> int:
        off = np.ascontiguousarray(clip_frame_offsets, np.int64)
        tot = C.c_size_t()
        _check(self._lib.af_stft_set_batch(self._h, _ptr(off), off.size - 1, C.byref(tot)))
        self.total_frames = tot.value
        return self.total_frames

    def power_device(self, pcm_ptr: int, power_ptr: int, stream: int = 0) -> None:
        _check(self._lib.af_stft_power_device(self._h, C.c_void_p(pcm_ptr), C.c_void_p(power_ptr),
                                              C.c_void_p(stream)))

    def power(self, pcm: np.ndarray) -> np.ndarray:
        """One clip (or the batch set before): rows of |X[k]|^2."""
        dt = np.int16 if self.cfg.sample_format == _ffi.AF_PCM_I16 else np.float32
        x = np.ascontiguousarray(pcm, dt).reshape(-1)
        if self.total_frames == 0:
            self.set_batch_uniform(1, x.size // self.cfg.channels)
        out = np.empty((self.total_frames, self.n_bins()), np.float32)
        _check(self._lib.af_stft_power(self._h, _ptr(x), x.size, _ptr(out), out.size))
        return out


class MelFilterbank:
    """``MelFilterbank::new(sr, n_fft, n_mels)`` [B]: weights and ``mel = power @ W.T`` on MFMA."""

    def __init__(self, sample_rate: int = 16000, n_fft: int = 1024, n_mels: int = 80,
                 f_min: float = 0.0, f_max: float = 0.0, mel_scale: int = _ffi.AF_MEL_HTK,
                 mel_norm: int = _ffi.AF_MEL_NORM_NONE, log_mel: bool = False, device: int = -1):
        self._lib = library()
        self.cfg = _ffi.MelConfig(sample_rate, n_fft, n_mels, f_min, f_max, mel_scale, mel_norm,
                                  int(log_mel), device)
        h = C.c_void_p()
        _check(self._lib.af_mel_create(C.byref(self.cfg), C.byref(h)))
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.af_mel_destroy(h)

    def weights(self) -> np.ndarray:
        W = np.empty((self.cfg.n_mels, self.cfg.n_fft // 2 + 1), np.float32)
        _check(self._lib.af_mel_weights(self._h, _ptr(W), W.size))
        return W

    def apply_device(self, power_ptr: int, n_frames: int, mel_ptr: int, stream: int = 0) -> None:
        _check(self._lib.af_mel_apply_device(self._h, C.c_void_p(power_ptr), n_frames,
                                             C.c_void_p(mel_ptr), C.c_void_p(stream)))

    def apply(self, power: np.ndarray) -> np.ndarray:
        p = np.ascontiguousarray(power, np.float32)
        n = p.shape[0] if p.ndim == 2 else p.size // (self.cfg.n_fft // 2 + 1)
        mel = np.empty((n, self.cfg.n_mels), np.float32)
        _check(self._lib.af_mel_apply(self._h, _ptr(p), n, _ptr(mel), mel.size))
        return mel


class ChainPipeline:
    """BASELINE config 4 chain [B]: to_mono -> resample -> FIR -> STFT -> mel on the device.
    ``sample_format``: AF_PCM_F32 or the wire format's AF_PCM_I16 (websocket.rs:246-251)."""

    def __init__(self, input_rate: int = 48000, channels: int = 2, output_rate: int = 16000,
                 fir_taps=None, hop: int = 256, n_mels: int = 80, f_min: float = 0.0,
                 f_max: float = 0.0, device: int = -1, sample_format: int = _ffi.AF_PCM_F32):
        self._lib = library()
        self._taps = _f32(fir_taps) if fir_taps is not None and len(fir_taps) else None
        self.cfg = _chain_config(self._taps, input_rate, channels, output_rate, hop, n_mels, f_min, f_max, device,
                                 sample_format)
        h = C.c_void_p()
        _check(self._lib.af_chain_create(C.byref(self.cfg), C.byref(h)))
        self._h = h
        self.total_frames = 0

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.af_chain_destroy(h)

    def _dtype(self):
        return np.int16 if self.cfg.sample_format == _ffi.AF_PCM_I16 else np.float32

    def n_frames(self, n_pcm_frames: int) -> int:
        return int(self._lib.af_chain_n_frames(self._h, n_pcm_frames))

    def set_batch_uniform(self, n_clips: int, frames_per_clip: int) -> int:
        tot = C.c_size_t()
        _check(self._lib.af_chain_set_batch_uniform(self._h, n_clips, frames_per_clip, C.byref(tot)))
        self.total_frames = tot.value
        return self.total_frames

    def set_batch(self, clip_offsets) -> int:
        """Ragged batch (AudioFrame is variable-length, capture.rs:13-28): clip i = PCM frames
        [clip_offsets[i], clip_offsets[i + 1]) of the buffer."""
        off = np.ascontiguousarray(clip_offsets, np.int64)
        tot = C.c_size_t()
        _check(self._lib.af_chain_set_batch(self._h, _ptr(off), off.size - 1, C.byref(tot)))
        self.total_frames = tot.value
        return self.total_frames

    def set_option(self, key: str, value: int) -> None:
        """Per-handle diagnostics: ``"unfused"`` 1 / 0 / -1 (follow af_debug_set)."""
        _check(self._lib.af_chain_set_option(self._h, key.encode(), int(value)))

    def run_device(self, pcm_ptr: int, mel_ptr: int, stream: int = 0) -> None:
        _check(self._lib.af_chain_run_device(self._h, C.c_void_p(pcm_ptr), C.c_void_p(mel_ptr),
                                             C.c_void_p(stream)))

    def n_launches(self) -> int:
        """Kernels one run_device() enqueues for the batch described (2 with the fused resampler + FIR stage)."""
        return int(self._lib.af_chain_n_launches(self._h))

    def run(self, pcm) -> np.ndarray:
        x = np.ascontiguousarray(pcm, self._dtype()).reshape(-1)
        mel = np.empty((self.total_frames, self.cfg.n_mels), np.float32)
        _check(self._lib.af_chain_run(self._h, _ptr(x), x.size, _ptr(mel), mel.size))
        return mel


class WireEncoder:
    """The reference's capture -> wire path on a batch of clips in ONE launch (``af_wire_*``): interleaved PCM ->
    ``AudioFrame::to_mono`` (capture.rs:30-42) -> ``AudioResampler::create_48k_to_16k`` / ``process`` (resampler.rs:60-93)
    -> ``(x.clamp(-1, 1) * 32767.0) as i16`` little-endian (websocket.rs:246-251) -> ``STANDARD.encode`` (websocket.rs:254).
    ``frame_samples = 0``: one base64 string per clip; ``frame_samples = 320``: one string per 20 ms send batch
    (specs/0002-design.md:1543-1546), a trailing partial batch is not sent."""

    def __init__(self, input_rate: int = 48000, output_rate: int = 16000, channels: int = 2,
                 sample_format: int = _ffi.AF_PCM_F32, frame_samples: int = 0, device: int = -1):
        self._lib = library()
        self.cfg = _ffi.WireConfig(input_rate, output_rate, channels, sample_format, frame_samples, device)
        h = C.c_void_p()
        _check(self._lib.af_wire_create(C.byref(self.cfg), C.byref(h)))
        self._h = h
        self.total_text = 0
        self.total_samples = 0
        self.n_clips = 0

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.af_wire_destroy(h)

    def _dtype(self):
        return np.int16 if self.cfg.sample_format == _ffi.AF_PCM_I16 else np.float32

    def out_samples(self, n_frames: int) -> int:
        return int(self._lib.af_wire_out_samples(self._h, n_frames))

    def text_len(self, n_frames: int) -> int:
        return int(self._lib.af_wire_text_len(self._h, n_frames))

    def message_len(self) -> int:
        return int(self._lib.af_wire_message_len(self._h))

    def n_messages(self, n_frames: int) -> int:
        return int(self._lib.af_wire_n_messages(self._h, n_frames))

    def set_batch_uniform(self, n_clips: int, frames_per_clip: int) -> int:
        tt, ts = C.c_size_t(), C.c_size_t()
        _check(self._lib.af_wire_set_batch_uniform(self._h, n_clips, frames_per_clip, C.byref(tt), C.byref(ts)))
        self.total_text, self.total_samples, self.n_clips = tt.value, ts.value, n_clips
        return self.total_text

    def set_batch(self, clip_offsets) -> int:
        off = np.ascontiguousarray(clip_offsets, np.int64)
        tt, ts = C.c_size_t(), C.c_size_t()
        _check(self._lib.af_wire_set_batch(self._h, _ptr(off), off.size - 1, C.byref(tt), C.byref(ts)))
        self.total_text, self.total_samples, self.n_clips = tt.value, ts.value, off.size - 1
        return self.total_text

    def text_offsets(self) -> np.ndarray:
        off = np.zeros(self.n_clips + 1, np.int64)
        _check(self._lib.af_wire_text_offsets(self._h, _ptr(off), off.size))
        return off

    def set_option(self, key: str, value: int) -> None:
        _check(self._lib.af_wire_set_option(self._h, key.encode(), int(value)))

    def n_launches(self) -> int:
        return int(self._lib.af_wire_n_launches(self._h))

    def total_messages(self) -> int:
        return int(self._lib.af_wire_total_messages(self._h))

    def run_device(self, pcm_ptr: int, text_ptr: int, pcm16_ptr: int = 0, stream: int = 0, energy_ptr: int = 0,
                   peak_ptr: int = 0) -> None:
        _check(self._lib.af_wire_run_device(self._h, C.c_void_p(pcm_ptr), C.c_void_p(text_ptr),
                                            C.c_void_p(pcm16_ptr) if pcm16_ptr else None,
                                            C.c_void_p(energy_ptr) if energy_ptr else None,
                                            C.c_void_p(peak_ptr) if peak_ptr else None, C.c_void_p(stream)))

    def run(self, pcm, with_pcm16: bool = False, with_level: bool = False):
        """Host buffers -> the batch's text as ``bytes`` (cut it with ``text_offsets()`` / ``message_len()``); with
        ``with_pcm16`` also the quantised samples (int16); with ``with_level`` also ``(energy, peak)`` per message --
        the mean-square energy ``VoiceActivityDetector::calculate_energy`` would compute on the message (vad.rs:157-168)
        and max |x| (the ``peak`` of the ``audio_level`` event, events/mod.rs:73-75)."""
        x = np.ascontiguousarray(pcm, self._dtype()).reshape(-1)
        text = np.empty(self.total_text, np.uint8)
        q = np.full(self.total_samples, -21846, np.int16) if with_pcm16 else None
        nm = self.total_messages() if with_level else 0
        en = np.full(nm, np.nan, np.float32) if with_level else None
        pk = np.full(nm, np.nan, np.float32) if with_level else None
        _check(self._lib.af_wire_run(self._h, _ptr(x), x.size, _ptr(text), text.size,
                                     _ptr(q) if q is not None else None, q.size if q is not None else 0,
                                     _ptr(en) if with_level else None, _ptr(pk) if with_level else None, nm))
        out = (text.tobytes(),)
        if with_pcm16:
            out += (q,)
        if with_level:
            out += (en, pk)
        return out if len(out) > 1 else out[0]

    def messages(self, pcm) -> List[List[str]]:
        """``frame_samples > 0``: per clip, the base64 payloads of its messages (what ``send_audio`` puts into
        ``audio_base_64``, websocket.rs:257-260)."""
        text = self.run(pcm)
        off = self.text_offsets()
        ml = self.message_len()
        out = []
        for i in range(self.n_clips):
            row = text[off[i]:off[i + 1]]
            out.append([row[k:k + ml].decode("ascii") for k in range(0, len(row), ml)] if ml else [row.decode("ascii")])
        return out


def _chain_config(taps, input_rate, channels, output_rate, hop, n_mels, f_min, f_max, device, sample_format):
    return _ffi.ChainConfig(input_rate, channels, output_rate, _ptr(taps) if taps is not None else None,
                            taps.size if taps is not None else 0, hop, n_mels, f_min, f_max, device, sample_format)


class ChainStream:
    """The whole chain over an unbounded stream (BASELINE config 5 on the full chain, ``af_chain_stream_*``):
    ``push`` returns the rows of every step it completed (one hipGraph replay each: H2D, fused resampler + FIR,
    STFT / mel, history move, D2H), ``flush`` the rows of the rest; all rows in order == ``ChainPipeline.run`` on
    the whole stream, bit for bit.  Carry / flush analogue in the reference: resampler.rs:132-166."""

    def __init__(self, step_frames: int = 1 << 20, input_rate: int = 48000, channels: int = 2,
                 output_rate: int = 16000, fir_taps=None, hop: int = 256, n_mels: int = 80, f_min: float = 0.0,
                 f_max: float = 0.0, device: int = -1, sample_format: int = _ffi.AF_PCM_F32):
        self._lib = library()
        self._taps = _f32(fir_taps) if fir_taps is not None and len(fir_taps) else None
        self.cfg = _chain_config(self._taps, input_rate, channels, output_rate, hop, n_mels, f_min, f_max, device,
                                 sample_format)
        h = C.c_void_p()
        _check(self._lib.af_chain_stream_create(C.byref(self.cfg), int(step_frames), C.byref(h)))
        self._h = h
        self.row_len = int(self._lib.af_chain_stream_row_len(h))
        self.rows_per_step = int(self._lib.af_chain_stream_rows_per_step(h))
        self.step_frames = int(self._lib.af_chain_stream_step_frames(h))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.af_chain_stream_destroy(h)

    def _dtype(self):
        return np.int16 if self.cfg.sample_format == _ffi.AF_PCM_I16 else np.float32

    def push(self, pcm) -> np.ndarray:
        x = np.ascontiguousarray(pcm, self._dtype()).reshape(-1)
        n_frames = x.size // self.cfg.channels
        rows = np.empty((int(self._lib.af_chain_stream_max_rows(self._h, n_frames)), self.row_len), np.float32)
        n = C.c_size_t()
        _check(self._lib.af_chain_stream_push(self._h, _ptr(x), n_frames, _ptr(rows), rows.size, C.byref(n)))
        return rows[:n.value]

    def flush(self) -> np.ndarray:
        # (the flush can owe one row more than a step when the resampler's look-ahead spans a hop)
        rows = np.empty((max(int(self._lib.af_chain_stream_flush_rows(self._h)), 1), self.row_len), np.float32)
        n = C.c_size_t()
        _check(self._lib.af_chain_stream_flush(self._h, _ptr(rows), rows.size, C.byref(n)))
        return rows[:n.value].copy()

    def reset(self) -> None:
        _check(self._lib.af_chain_stream_reset(self._h))

    def total_rows(self, n_frames: int) -> int:
        return int(self._lib.af_chain_stream_total_rows(self._h, int(n_frames)))

    @property
    def first_step_frames(self) -> int:
        return int(self._lib.af_chain_stream_first_step_frames(self._h))

    @property
    def history_frames(self) -> int:
        return int(self._lib.af_chain_stream_history_frames(self._h))

    def seek(self, step: int) -> int:
        """Time sharding (shard.chain_stream_shards): continue as if ``step`` steps had already run.  Returns the
        frame of the recording the next ``push`` must start at (the step's history comes first and yields no rows)."""
        first = C.c_uint64()
        _check(self._lib.af_chain_stream_seek(self._h, int(step), C.byref(first)))
        return int(first.value)

    def stats(self) -> dict:
        v = (C.c_double * 4)()
        _check(self._lib.af_chain_stream_stats(self._h, v))
        return {"device_ms": v[0], "wall_ms": v[1], "steps": int(v[2]), "rows": int(v[3])}
