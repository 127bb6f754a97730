"""ctypes view of oracle/libaf_oracle.so (the CPU checker).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libaf_oracle.so")


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB_PATH


def _load() -> C.CDLL:
    build()
    try:
        return C.CDLL(_LIB_PATH)
    except OSError:
        build(force=True)          # e.g. a stale .so from another host
        return C.CDLL(_LIB_PATH)


_lib = _load()

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i16p = np.ctypeslib.ndpointer(np.int16, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


class VadConfig(C.Structure):
    _fields_ = [("threshold_db", C.c_float), ("smoothing_factor", C.c_float),
                ("silence_timeout_frames", C.c_size_t), ("min_speech_frames", C.c_size_t)]


class _Vad(C.Structure):
    _fields_ = [("cfg", VadConfig), ("smoothed_energy", C.c_float),
                ("silence_frames", C.c_size_t), ("speech_frames", C.c_size_t),
                ("state", C.c_int)]


class _Ring(C.Structure):
    _fields_ = [("buf", C.POINTER(C.c_float)), ("capacity", C.c_size_t),
                ("write_pos", C.c_size_t), ("read_pos", C.c_size_t)]


def _sig(name, restype, *argtypes):
    fn = getattr(_lib, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)
    return fn


_to_mono = _sig("afo_to_mono", C.c_size_t, _f32p, C.c_size_t, C.c_uint, _f32p)
_quant = _sig("afo_quantize_pcm16", None, _f32p, C.c_size_t, _i16p)
_b64 = _sig("afo_base64_encode", C.c_size_t, _u8p, C.c_size_t, C.c_char_p)
_b64len = _sig("afo_base64_len", C.c_size_t, C.c_size_t)
_energy = _sig("afo_vad_energy", C.c_float, _f32p, C.c_size_t)
_dbfs = _sig("afo_energy_to_dbfs", C.c_float, C.c_float)
_vad_defcfg = _sig("afo_vad_default_config", None, C.POINTER(VadConfig))
_vad_init = _sig("afo_vad_init", None, C.POINTER(_Vad), C.POINTER(VadConfig))
_vad_detect = _sig("afo_vad_detect", C.c_int, C.POINTER(_Vad), _f32p, C.c_size_t)
_vad_step = _sig("afo_vad_step_energy", C.c_int, C.POINTER(_Vad), C.c_float)
_vad_reset = _sig("afo_vad_reset", None, C.POINTER(_Vad))
_vad_scan = _sig("afo_vad_scan_batch", None, C.POINTER(VadConfig), _f32p, C.c_size_t, C.c_size_t, _u8p)
_ring_init = _sig("afo_ring_init", C.c_int, C.POINTER(_Ring), C.c_size_t)
_ring_free = _sig("afo_ring_free", None, C.POINTER(_Ring))
_ring_write = _sig("afo_ring_write", C.c_size_t, C.POINTER(_Ring), _f32p, C.c_size_t)
_ring_read = _sig("afo_ring_read", C.c_size_t, C.POINTER(_Ring), C.c_size_t, _f32p)
_ring_avail = _sig("afo_ring_available", C.c_size_t, C.POINTER(_Ring))
_ring_clear = _sig("afo_ring_clear", None, C.POINTER(_Ring))
_chunker = _sig("afo_chunker_process", None, C.c_size_t, C.c_size_t, C.c_size_t,
                C.POINTER(C.c_size_t), C.POINTER(C.c_size_t))
_ratio = _sig("afo_resample_ratio", None, C.c_uint32, C.c_uint32,
              C.POINTER(C.c_uint32), C.POINTER(C.c_uint32))
_outlen = _sig("afo_resample_out_len", C.c_size_t, C.c_size_t, C.c_uint32, C.c_uint32)
_design = _sig("afo_resample_design", None, C.c_uint32, C.c_uint32, C.c_uint32, _f64p)
_res64 = _sig("afo_resample_f64", None, _f32p, C.c_size_t, _f64p, C.c_uint32, C.c_uint32,
              C.c_uint32, C.c_size_t, C.c_size_t, _f64p)
_res32 = _sig("afo_resample_f32", None, _f32p, C.c_size_t, _f32p, C.c_uint32, C.c_uint32,
              C.c_uint32, C.c_size_t, C.c_size_t, _f32p)
_fir64 = _sig("afo_fir_f64", None, _f32p, C.c_size_t, _f32p, C.c_size_t, _f64p)
_fir32 = _sig("afo_fir_f32", None, _f32p, C.c_size_t, _f32p, C.c_size_t, _f32p)
_level = _sig("afo_level_meter_f64", None, _f32p, C.c_size_t, C.POINTER(C.c_double),
             C.POINTER(C.c_double), C.POINTER(C.c_double))
_nframes = _sig("afo_stft_n_frames", C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t)
_hann = _sig("afo_hann_periodic", None, C.c_size_t, _f64p)
_stft = _sig("afo_stft_power_f64", None, _f32p, C.c_size_t, C.c_size_t, C.c_size_t, _f64p)
_dft = _sig("afo_dft_power_naive_f64", None, _f32p, C.c_size_t, _f64p, _f64p)
_hz2mel = _sig("afo_hz_to_mel_htk", C.c_double, C.c_double)
_mel2hz = _sig("afo_mel_to_hz_htk", C.c_double, C.c_double)
_melw = _sig("afo_mel_weights", None, C.c_double, C.c_size_t, C.c_size_t, C.c_double,
             C.c_double, _f64p)
_melw_ex = _sig("afo_mel_weights_ex", None, C.c_double, C.c_size_t, C.c_size_t, C.c_double,
                C.c_double, C.c_int, C.c_int, _f64p)
_melapply = _sig("afo_mel_apply_f64", None, _f64p, C.c_size_t, C.c_size_t, _f64p,
                 C.c_size_t, _f64p)
_feat64 = _sig("afo_features_f64", C.c_size_t, _f32p, C.c_size_t, C.c_uint, C.c_double,
               C.c_size_t, C.c_size_t, C.c_size_t, C.c_double, C.c_double, _f64p)
_feat32 = _sig("afo_features_f32_batch", C.c_size_t, _f32p, C.c_size_t, C.c_size_t, C.c_uint,
               C.c_double, C.c_size_t, C.c_size_t, C.c_size_t, C.c_double, C.c_double,
               _f32p, C.c_int)
_maxthreads = _sig("afo_max_threads", C.c_int)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


# ---- [R] -------------------------------------------------------------------
def to_mono(samples, channels: int) -> np.ndarray:
    x = _f32(samples).ravel()
    if channels == 0:
        raise ValueError("channels == 0")
    out = np.empty((x.size + channels - 1) // channels if channels > 1 else x.size, np.float32)
    n = _to_mono(x, x.size, channels, out)
    return out[:n]


def quantize_pcm16(samples) -> np.ndarray:
    x = _f32(samples).ravel()
    out = np.empty(x.size, np.int16)
    _quant(x, x.size, out)
    return out


def base64_encode(data: bytes) -> str:
    a = np.frombuffer(data, np.uint8) if len(data) else np.zeros(0, np.uint8)
    buf = C.create_string_buffer(int(_b64len(len(data))) + 1)
    n = _b64(np.ascontiguousarray(a), len(data), buf)
    return buf.raw[:n].decode("ascii")


def vad_energy(frame) -> float:
    x = _f32(frame).ravel()
    return float(_energy(x, x.size))


def energy_to_dbfs(e: float) -> float:
    return float(_dbfs(float(e)))


SILENCE, SPEECH, ENDING = 0, 1, 2


def vad_scan_batch(energy, threshold_db=-50.0, smoothing_factor=0.3, silence_timeout_frames=15,
                   min_speech_frames=3) -> np.ndarray:
    """n_clips independent detectors (vad.rs:60-70) over energy[n_clips][n_frames] -> states uint8."""
    e = np.ascontiguousarray(energy, dtype=np.float32)
    assert e.ndim == 2
    cfg = VadConfig(threshold_db, smoothing_factor, silence_timeout_frames, min_speech_frames)
    out = np.zeros(e.shape, np.uint8)
    _vad_scan(C.byref(cfg), e.reshape(-1), e.shape[0], e.shape[1], out.reshape(-1))
    return out


class Vad:
    def __init__(self, threshold_db=-50.0, smoothing_factor=0.3,
                 silence_timeout_frames=15, min_speech_frames=3):
        self._cfg = VadConfig(threshold_db, smoothing_factor, silence_timeout_frames,
                              min_speech_frames)
        self._v = _Vad()
        _vad_init(C.byref(self._v), C.byref(self._cfg))

    def detect(self, frame) -> int:
        x = _f32(frame).ravel()
        return int(_vad_detect(C.byref(self._v), x, x.size))

    def step_energy(self, e: float) -> int:
        return int(_vad_step(C.byref(self._v), float(e)))

    def reset(self):
        _vad_reset(C.byref(self._v))

    @property
    def state(self) -> int:
        return int(self._v.state)

    @property
    def smoothed_energy(self) -> float:
        return float(self._v.smoothed_energy)

    @property
    def speech_frames(self) -> int:
        return int(self._v.speech_frames)


class Ring:
    def __init__(self, capacity: int):
        self._r = _Ring()
        if _ring_init(C.byref(self._r), capacity) != 0:
            raise MemoryError
        self.capacity = capacity

    def __del__(self):
        try:
            _ring_free(C.byref(self._r))
        except Exception:
            pass

    def write(self, data) -> int:
        x = _f32(data).ravel()
        return int(_ring_write(C.byref(self._r), x, x.size))

    def read(self, size: int):
        out = np.empty(max(size, 1), np.float32)
        n = _ring_read(C.byref(self._r), size, out)
        if n == C.c_size_t(-1).value:
            return None
        return out[:n].copy()

    def available(self) -> int:
        return int(_ring_avail(C.byref(self._r)))

    def clear(self):
        _ring_clear(C.byref(self._r))


def chunker_process(carry_in: int, n_in: int, chunk: int):
    a, b = C.c_size_t(), C.c_size_t()
    _chunker(carry_in, n_in, chunk, C.byref(a), C.byref(b))
    return int(a.value), int(b.value)


# ---- [B] -------------------------------------------------------------------
def resample_ratio(in_hz: int, out_hz: int):
    L, M = C.c_uint32(), C.c_uint32()
    _ratio(in_hz, out_hz, C.byref(L), C.byref(M))
    return int(L.value), int(M.value)


def resample_out_len(n_in: int, L: int, M: int) -> int:
    return int(_outlen(n_in, L, M))


def resample_design(L: int, M: int, T: int = 64) -> np.ndarray:
    h = np.empty(L * T, np.float64)
    _design(L, M, T, h)
    return h


def resample(x, in_hz: int, out_hz: int, T: int = 64, m0: int = 0, n_out=None) -> np.ndarray:
    x = _f32(x).ravel()
    L, M = resample_ratio(in_hz, out_hz)
    h = resample_design(L, M, T)
    if n_out is None:
        n_out = resample_out_len(x.size, L, M) - m0
    y = np.empty(n_out, np.float64)
    _res64(x, x.size, h, L, M, T, m0, n_out, y)
    return y


_cubic = _sig("afo_resample_cubic", None, _f32p, C.c_size_t, C.c_uint, C.c_uint, C.c_size_t,
              np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS"))


def resample_cubic(x, in_hz: int, out_hz: int) -> np.ndarray:
    """docs/OPS.md "Cubic interpolation" (the mode resampler.rs:43-49 configures), float64."""
    x = _f32(x).ravel()
    L, M = resample_ratio(in_hz, out_hz)
    y = np.empty(resample_out_len(x.size, L, M), np.float64)
    _cubic(x, x.size, in_hz, out_hz, y.size, y)
    return y


def resample_f32(x, in_hz: int, out_hz: int, T: int = 64) -> np.ndarray:
    x = _f32(x).ravel()
    L, M = resample_ratio(in_hz, out_hz)
    h = resample_design(L, M, T).astype(np.float32)
    n_out = resample_out_len(x.size, L, M)
    y = np.empty(n_out, np.float32)
    _res32(x, x.size, h, L, M, T, 0, n_out, y)
    return y


def fir(x, h) -> np.ndarray:
    x = _f32(x).ravel()
    h = _f32(h).ravel()
    y = np.empty(x.size, np.float64)
    _fir64(x, x.size, h, h.size, y)
    return y


def fir_f32(x, h) -> np.ndarray:
    x = _f32(x).ravel()
    h = _f32(h).ravel()
    y = np.empty(x.size, np.float32)
    _fir32(x, x.size, h, h.size, y)
    return y


def level_meter(window):
    x = _f32(window).ravel()
    r, p, d = C.c_double(), C.c_double(), C.c_double()
    _level(x, x.size, C.byref(r), C.byref(p), C.byref(d))
    return r.value, p.value, d.value


def stft_n_frames(n: int, n_fft: int = 1024, hop: int = 256) -> int:
    return int(_nframes(n, n_fft, hop))


def hann_periodic(N: int) -> np.ndarray:
    w = np.empty(N, np.float64)
    _hann(N, w)
    return w


def stft_power(x, n_fft: int = 1024, hop: int = 256) -> np.ndarray:
    x = _f32(x).ravel()
    nf = stft_n_frames(x.size, n_fft, hop)
    p = np.empty((nf, n_fft // 2 + 1), np.float64)
    if nf:
        _stft(x, x.size, n_fft, hop, p)
    return p


def dft_power_naive(frame, n_fft: int) -> np.ndarray:
    x = _f32(frame).ravel()
    w = hann_periodic(n_fft)
    p = np.empty(n_fft // 2 + 1, np.float64)
    _dft(x, n_fft, w, p)
    return p


def hz_to_mel(hz: float) -> float:
    return float(_hz2mel(hz))


def mel_to_hz(mel: float) -> float:
    return float(_mel2hz(mel))


def mel_weights(sample_rate: float, n_fft: int = 1024, n_mels: int = 80,
                f_min: float = 0.0, f_max=None, scale: int = 0, norm: int = 0) -> np.ndarray:
    """scale 0 HTK / 1 Slaney; norm 1 = Slaney area normalisation (docs/OPS.md)."""
    if f_max is None:
        f_max = sample_rate / 2.0
    W = np.empty((n_mels, n_fft // 2 + 1), np.float64)
    _melw_ex(sample_rate, n_fft, n_mels, f_min, f_max, scale, norm, W)
    return W


def mel_apply(power: np.ndarray, W: np.ndarray) -> np.ndarray:
    power = np.ascontiguousarray(power, np.float64)
    W = np.ascontiguousarray(W, np.float64)
    out = np.empty((power.shape[0], W.shape[0]), np.float64)
    _melapply(power, power.shape[0], power.shape[1], W, W.shape[0], out)
    return out


def features(pcm, channels: int, sample_rate: float, n_fft: int = 1024, hop: int = 256,
             n_mels: int = 80, f_min: float = 0.0, f_max=None) -> np.ndarray:
    """One clip: interleaved PCM -> to_mono -> STFT power -> mel, float64."""
    x = _f32(pcm).ravel()
    if f_max is None:
        f_max = sample_rate / 2.0
    n_frames_pcm = x.size // channels
    nf = stft_n_frames(n_frames_pcm, n_fft, hop)
    mel = np.empty((nf, n_mels), np.float64)
    if nf:
        _feat64(x, n_frames_pcm, channels, sample_rate, n_fft, hop, n_mels, f_min, f_max, mel)
    return mel


def features_f32_batch(pcm, n_clips: int, n_frames_pcm: int, channels: int, sample_rate: float,
                       n_fft: int = 1024, hop: int = 256, n_mels: int = 80,
                       f_min: float = 0.0, f_max=None, threads: int = 1) -> np.ndarray:
    x = _f32(pcm).ravel()
    if f_max is None:
        f_max = sample_rate / 2.0
    nf = stft_n_frames(n_frames_pcm, n_fft, hop)
    mel = np.empty((n_clips, nf, n_mels), np.float32)
    if nf:
        _feat32(x, n_clips, n_frames_pcm, channels, sample_rate, n_fft, hop, n_mels,
                f_min, f_max, mel, threads)
    return mel


def max_threads() -> int:
    return int(_maxthreads())
