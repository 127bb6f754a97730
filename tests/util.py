"""Shared helpers for the tests: deterministic synthetic PCM (BASELINE.md:
three sines + +-0.01 uniform noise, seed 0xA5D10F10 + clip index) and error metrics."""
import numpy as np

SEED0 = 0xA5D10F10


def synth_clip(clip_index: int, n_frames: int, channels: int, sample_rate: int) -> np.ndarray:
    """Interleaved f32 PCM in [-1, 1], shape (n_frames * channels,)."""
    rng = np.random.default_rng(SEED0 + clip_index)
    t = np.arange(n_frames, dtype=np.float64) / sample_rate
    x = np.zeros((n_frames, channels))
    for ch in range(channels):
        for _ in range(3):
            f = np.exp(rng.uniform(np.log(50.0), np.log(0.45 * sample_rate)))
            a = rng.uniform(0.05, 0.3)
            ph = rng.uniform(0, 2 * np.pi)
            x[:, ch] += a * np.sin(2 * np.pi * f * t + ph)
    x += rng.uniform(-0.01, 0.01, x.shape)
    return np.clip(x, -1.0, 1.0).astype(np.float32).reshape(-1)


def frame_rel_err(got: np.ndarray, ref: np.ndarray) -> float:
    """max over rows of ||got - ref||_2 / ||ref||_2 (docs/OPS.md tolerance definition)."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    if ref.size == 0:
        return 0.0
    num = np.linalg.norm(got - ref, axis=-1)
    den = np.maximum(np.linalg.norm(ref, axis=-1), 1e-300)
    return float(np.max(num / den))


def rel_err(got, ref) -> float:
    got = np.asarray(got, np.float64).ravel()
    ref = np.asarray(ref, np.float64).ravel()
    return float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-300))
