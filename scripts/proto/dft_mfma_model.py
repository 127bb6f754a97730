"""Numpy model of the MFMA formulation of the hop-256 feature kernel (features_v3.hip).

Checks, on the CPU, (1) the index algebra of the 32 x 32 two-stage DFT with every twiddle and the
window folded into the stage-1 matrices -- row order, scales and bin mapping as in the kernel --
and (2) the accuracy of the f16-pair operands (x = hi + lo' / 2048; three products per MAC:
hi*hi -> P1, hi*lo' + lo'*hi -> P2, f32 accumulation, value P1 + P2 / 2048) against a float64 rfft.
tests/test_dft_mfma_model.py runs it; the kernel itself is checked on the GPU (tests/test_gpu_features_v3.py).
"""
import numpy as np

N = 1024
HOP = 256


def hann(n_fft=N):
    w = np.zeros(N)
    i = np.arange(n_fft)
    w[:n_fft] = 0.5 - 0.5 * np.cos(2 * np.pi * i / n_fft)
    return w


def stage1_matrix(n2, w, scale=1.0 / 64.0):
    """A1[n2]: 32 rows x 32 (n1).  Rows 0..15: Re k1 = 0..15; rows 16..31: row 16 = Re k1 = 16 (untwiddled), 17..31 = Im k1 = 1..15."""
    n1 = np.arange(32)
    n = 32 * n1 + n2
    A = np.zeros((32, 32))
    for m in range(16):
        A[m] = w[n] * np.cos(2 * np.pi * n * m / N)
    A[16] = w[n] * np.cos(np.pi * n1)          # (-1)^n1
    for m in range(1, 16):
        A[16 + m] = -w[n] * np.sin(2 * np.pi * n * m / N)
    return A * scale


def k2_of(rb, m):
    """row m of row block rb: q = m >> 2, i = m & 3  ->  k2 = 8 rb + 4 (q >> 1) + (q & 1) + 2 (i >> 1)  (features_v3.hip)"""
    q, i = m >> 2, m & 3
    return 8 * rb + 4 * (q >> 1) + (q & 1) + 2 * (i >> 1)


def stage2_matrix(scale=1.0):
    """A2: 64 rows x 32 (n2).  Row 16 rb + m is k2_of(rb, m); type m & 1 (0 cos, 1 -sin)."""
    n2 = np.arange(32)
    A = np.zeros((64, 32))
    for r in range(64):
        rb, m = divmod(r, 16)
        k2 = k2_of(rb, m)
        ang = 2 * np.pi * n2 * k2 / 32
        A[r] = -np.sin(ang) if (m & 1) else np.cos(ang)
    return A * scale


def stage2_special(scale=1.0):
    """A2s (k1 = 16): 32 rows (k2 = 0..15) x 32 (n2), the twiddle W64^n2 folded in."""
    n2 = np.arange(32)
    A = np.zeros((32, 32))
    for r in range(32):
        rb, m = divmod(r, 16)
        k2 = k2_of(rb, m)
        ang = 2 * np.pi * n2 * (2 * k2 + 1) / 64
        A[r] = -np.sin(ang) if (m & 1) else np.cos(ang)
    return A * scale


def split16(a):
    """x -> (hi, lo') with hi = f16(x), lo' = f16((x - hi) 2048)"""
    hi = a.astype(np.float16)
    lo = ((a - hi.astype(np.float64)) * 2048.0).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)


def mm3(A, B, exact):
    """A @ B the way the kernel does it: P1 = hi hi, P2 = hi lo' + lo' hi (f32 accumulation), value P1 + P2 / 2048."""
    if exact:
        return A @ B
    ah, al = split16(A)
    bh, bl = split16(B)
    p1 = (ah @ bh).astype(np.float32)
    p2 = (ah @ bl + al @ bh).astype(np.float32)
    return (p1 + p2 * np.float32(1.0 / 2048.0)).astype(np.float32)


def power_rows(x, n_frames, exact=True, n_fft=N):
    """x: mono samples of one 16-frame unit (>= 15 * HOP + N).  Returns power[n_frames][513]."""
    w = hann(n_fft)
    # block scale: max |x| -> [2^13, 2^14)
    mx = np.max(np.abs(x))
    e = 0 if mx == 0 else 13 - int(np.floor(np.log2(mx)))
    xs = x * 2.0 ** e
    if not exact:
        xs = xs.astype(np.float32).astype(np.float64)
    F = n_frames
    # stage 1: for every n2: D[32 rows][F] = A1[n2] (32 x 32) @ B[n1][f], B = xs[256 f + 32 n1 + n2]
    Y = np.zeros((32, 32, F))       # [row][n2][f]
    f = np.arange(F)
    n1 = np.arange(32)
    for n2 in range(32):
        B = xs[(HOP * f[None, :] + 32 * n1[:, None] + n2)]
        Y[:, n2, :] = mm3(stage1_matrix(n2, w), B, exact)
    A2 = stage2_matrix()
    A2s = stage2_special()
    P = np.zeros((F, 513))
    inv = (64.0 * 2.0 ** -e) ** 2                 # X_acc = X s / 64 with s = 2^e: the mel rows are scaled by (64 / s)^2
    for k1 in range(1, 16):
        Yr, Yi = Y[k1], Y[16 + k1]               # [n2][f]
        a0 = mm3(A2, Yr, exact)
        a1 = mm3(A2, Yi, exact)
        for r in range(0, 64, 2):
            k2 = k2_of(*divmod(r, 16))
            re = a0[r] - a1[r + 1]
            im = a1[r] + a0[r + 1]
            b = k1 + 32 * k2 if k2 < 16 else (32 - k1) + 32 * (31 - k2)
            P[:, b] = (re * re + im * im) * inv
    # special: k1 = 0 (rows of A2 for k2 = 0..16), k1 = 16 (A2s)
    a0 = mm3(A2[:48], Y[0], exact)
    for r in range(0, 48, 2):
        k2 = k2_of(*divmod(r, 16))
        if k2 <= 16:
            P[:, 32 * k2] = (a0[r] ** 2 + a0[r + 1] ** 2) * inv
    a16 = mm3(A2s, Y[16], exact)
    for r in range(0, 32, 2):
        k2 = k2_of(*divmod(r, 16))
        P[:, 16 + 32 * k2] = (a16[r] ** 2 + a16[r + 1] ** 2) * inv
    return P


def reference(x, n_frames, n_fft=N):
    w = hann(n_fft)[:n_fft]
    out = np.zeros((n_frames, n_fft // 2 + 1))
    for f in range(n_frames):
        out[f] = np.abs(np.fft.rfft(x[HOP * f:HOP * f + n_fft] * w)) ** 2
    return out


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    L = 15 * HOP + N
    for name, x in [
        ("noise", rng.standard_normal(L) * 0.3),
        ("tone+noise", np.sin(2 * np.pi * 440.0 / 48000 * np.arange(L)) + 1e-3 * rng.standard_normal(L)),
        ("quiet tail", np.concatenate([rng.standard_normal(L // 2), 1e-4 * rng.standard_normal(L - L // 2)])),
        ("very quiet tail", np.concatenate([rng.standard_normal(L // 2), 1e-6 * rng.standard_normal(L - L // 2)])),
        ("dc", np.full(L, 0.9)),
        ("tiny", 1e-12 * rng.standard_normal(L)),
    ]:
        ref = reference(x, 16)
        ex = power_rows(x, 16, exact=True)
        ap = power_rows(x, 16, exact=False)
        e_exact = np.max(np.linalg.norm(ex - ref, axis=1) / np.linalg.norm(ref, axis=1))
        e_split = np.linalg.norm(ap - ref, axis=1) / np.linalg.norm(ref, axis=1)
        print(f"{name:16s} algebra {e_exact:.2e}   f16-split per-frame rel err: max {e_split.max():.2e} median {np.median(e_split):.2e}")
    # n_fft 512: bins 0, 2, 4 ... of the 1024-point transform
    x = rng.standard_normal(L)
    ref = reference(x, 16, 512)
    ex = power_rows(x, 16, True, 512)[:, ::2]
    print("n_fft 512 algebra", np.max(np.abs(ex - ref) / np.max(ref)))
