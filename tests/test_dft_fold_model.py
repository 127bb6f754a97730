"""Host model of the matrix-product DFT kernel's algebra (features_dft.hip / build_dft in features_api.hip): the frame folded on the
symmetry of the periodic Hann window, the tables in the kernel's (bin tile, K group, lane, step) order and the K-slot mapping of
`v_mfma_f32_16x16x4_f32`, in float64 -- against numpy's rfft.  No GPU; what the GPU tests add is the f32 arithmetic."""
import numpy as np
import pytest


def tables(n_fft):
    """tfrag[bt][g][c][lane][j] as build_dft writes them (float64 here)"""
    n_bins, nh = n_fft // 2 + 1, n_fft // 2
    kg, nb = max((nh + 15) // 16, 4), (n_bins + 15) // 16
    win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)
    t = np.zeros((nb, kg, 2, 64, 4))
    for bt in range(nb):
        for g in range(kg):
            for lane in range(64):
                for j in range(4):
                    k, b = 16 * g + 4 * (lane >> 4) + j, 16 * bt + (lane & 15)
                    if k >= nh or b >= n_bins:
                        continue
                    nn = k + 1
                    r = (nn * b) % n_fft
                    t[bt, g, 0, lane, j] = win[nn] * np.cos(2 * np.pi * r / n_fft)
                    t[bt, g, 1, lane, j] = -win[nn] * np.sin(2 * np.pi * r / n_fft)
    return t, kg, nb


def folded_rows(frames, n_fft, kg):
    """E, O [frame][16 kg] as the kernel builds them"""
    nh, kpad = n_fft // 2, 16 * kg
    E, O = np.zeros((frames.shape[0], kpad)), np.zeros((frames.shape[0], kpad))
    for k in range(nh):
        nn = k + 1
        if n_fft - nn == nn:
            E[:, k] = frames[:, nn]
        else:
            E[:, k] = frames[:, nn] + frames[:, n_fft - nn]
            O[:, k] = frames[:, nn] - frames[:, n_fft - nn]
    return E, O


def mfma_16x16x4(a_lane, b_lane, acc):
    """D[4 (lane >> 4) + i][lane & 15] += sum_kq A[m][kq] B[kq][n]; A: lane (m = lane & 15, kq = lane >> 4), B: lane (n, kq)"""
    A = np.zeros((16, 4)); B = np.zeros((4, 16))
    for lane in range(64):
        A[lane & 15, lane >> 4] = a_lane[lane]
        B[lane >> 4, lane & 15] = b_lane[lane]
    return acc + A @ B


@pytest.mark.parametrize("n_fft", [400, 33, 100, 480, 1000])
def test_folded_matrix_product_equals_the_dft(n_fft):
    rng = np.random.default_rng(n_fft)
    frames = rng.standard_normal((16, n_fft))
    t, kg, nb = tables(n_fft)
    E, O = folded_rows(frames, n_fft, kg)
    n_bins = n_fft // 2 + 1
    win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)
    ref = np.abs(np.fft.rfft(frames * win, axis=1)) ** 2                      # [frame][bin]
    got = np.zeros((16, 16 * nb))
    lanes = np.arange(64)
    for bt in range(nb):
        are, aim = np.zeros((16, 16)), np.zeros((16, 16))                   # D[bin in tile][frame]
        for g in range(kg):
            for j in range(4):
                # B operand: lane (frame n = lane & 15, kq) supplies E / O [n][16 g + 4 kq + j]
                be = E[lanes & 15, 16 * g + 4 * (lanes >> 4) + j]
                bo = O[lanes & 15, 16 * g + 4 * (lanes >> 4) + j]
                are = mfma_16x16x4(t[bt, g, 0, :, j], be, are)
                aim = mfma_16x16x4(t[bt, g, 1, :, j], bo, aim)
        got[:, 16 * bt:16 * bt + 16] = (are ** 2 + aim ** 2).T
    err = np.abs(got[:, :n_bins] - ref).max() / ref.max()
    assert err < 1e-12
    assert np.all(got[:, n_bins:] == 0.0)                                    # the padding bins of the last tile
