"""Index-map check of the v2 feature kernel's FFT: 1024 = 16 (regs) x 16 (regs) x 4 (regs) with two
LDS transpositions; lanes 0..63.  Mirrors features_v2.hip stage by stage (numpy, complex128)."""
import numpy as np
rng = np.random.default_rng(0)
N = 1024
z = rng.standard_normal(N) + 1j * rng.standard_normal(N)
W = lambda n, e: np.exp(-2j * np.pi * (e % n) / n)

# stage 1: lane n2, regs n1
Y = np.zeros((16, 64), complex)            # [k1][n2]
for n2 in range(64):
    x = np.array([z[64 * n1 + n2] for n1 in range(16)])
    y = np.fft.fft(x)                      # radix-16 over n1
    for k1 in range(16):
        Y[k1, n2] = y[k1] * W(1024, n2 * k1)
# T1: reader lane L1 = k1 + 16 c reads row k1, cols c + 4 m
U = np.zeros((16, 64), complex)            # plane b, col L1
for L1 in range(64):
    k1, c = L1 & 15, L1 >> 4
    x = np.array([Y[k1, c + 4 * m] for m in range(16)])
    u = np.fft.fft(x)
    for b in range(16):
        U[b, L1] = u[b] * W(64, c * b)
# T2: reader lane L2 = k1 + 16 j reads planes j + 4 b', cols k1 + 16 c
Z = np.zeros((64, 16), complex)            # lane L2, reg t = b' + 4 a
for L2 in range(64):
    k1, j = L2 & 15, L2 >> 4
    for bp in range(4):
        x = np.array([U[j + 4 * bp, k1 + 16 * c] for c in range(4)])
        X = np.fft.fft(x)                  # radix-4 over c -> a
        for a in range(4):
            Z[L2, bp + 4 * a] = X[a]
ref = np.fft.fft(z)
got = np.zeros(N, complex)
for L2 in range(64):
    for t in range(16):
        got[L2 + 64 * t] = Z[L2, t]
print("fft err", np.abs(got - ref).max())
# split of two real frames packed as z = a + i b
a = rng.standard_normal(N); b = rng.standard_normal(N)
z = a + 1j * b
Zf = np.fft.fft(z)
Zl = np.array([[Zf[L + 64 * t] for t in range(16)] for L in range(64)])
PA = np.zeros(513); PB = np.zeros(513)
for L in range(64):
    d = 1 if L == 0 else 0
    pl = (64 - L) % 64
    for t in range(9):
        k = L + 64 * t
        if k > 512: continue
        idx = 15 - t + d
        P = Zl[pl, idx] if idx < 16 else Zl[pl, 0]       # plane 16 := own Z[0] (lane 0 only)
        zr, zi, pr, pi = Zl[L, t].real, Zl[L, t].imag, P.real, P.imag
        PA[k] = ((zr + pr) ** 2 + (zi - pi) ** 2) / 4
        PB[k] = ((zi + pi) ** 2 + (zr - pr) ** 2) / 4
print("split err", np.abs(PA - np.abs(np.fft.rfft(a)) ** 2).max(), np.abs(PB - np.abs(np.fft.rfft(b)) ** 2).max())
