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


def bits_equal(a, b) -> bool:
    """True iff `a` and `b` have the same shape and the same BITS: floats are compared through their integer images,
    so +0.0 != -0.0 and two NaNs are equal only with equal payloads.  This is what "bit-exact" / "bit for bit" means in
    the tests (np.array_equal / torch.equal compare values: 0.0 == -0.0, NaN != NaN).  Integers compare as they are.
    Accepts numpy arrays, torch tensors (compared where they live) and anything np.asarray takes; both sides must have
    the same element type."""
    try:
        import torch
    except ImportError:                      # pragma: no cover
        torch = None
    if torch is not None and (isinstance(a, torch.Tensor) or isinstance(b, torch.Tensor)):
        if not isinstance(a, torch.Tensor):
            a = torch.from_numpy(np.ascontiguousarray(a))
        if not isinstance(b, torch.Tensor):
            b = torch.from_numpy(np.ascontiguousarray(b))
        if a.device != b.device:
            a, b = a.cpu(), b.cpu()
        assert a.dtype == b.dtype, f"bits_equal: element types differ ({a.dtype} vs {b.dtype})"
        if a.shape != b.shape:
            return False
        if a.dtype.is_floating_point:
            image = {2: torch.int16, 4: torch.int32, 8: torch.int64}[a.element_size()]
            a, b = a.contiguous().view(image), b.contiguous().view(image)
        return bool(torch.equal(a, b))
    # a plain list / scalar takes the element type of the array it is compared with
    if not isinstance(a, np.ndarray) and isinstance(b, np.ndarray):
        a = np.asarray(a, dtype=b.dtype)
    if not isinstance(b, np.ndarray) and isinstance(a, np.ndarray):
        b = np.asarray(b, dtype=a.dtype)
    a = np.ascontiguousarray(np.asarray(a))
    b = np.ascontiguousarray(np.asarray(b))
    assert a.dtype == b.dtype, f"bits_equal: element types differ ({a.dtype} vs {b.dtype})"
    if a.shape != b.shape:
        return False
    if a.dtype.kind == "f":
        image = {2: np.uint16, 4: np.uint32, 8: np.uint64}[a.dtype.itemsize]
        a, b = a.view(image), b.view(image)
    return bool(np.array_equal(a, b))
