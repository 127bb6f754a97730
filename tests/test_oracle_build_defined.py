"""The [B] half of the oracle (docs/OPS.md) against independent numpy/scipy
formulations, the committed golden vectors, and its own float32 port (the
timed cpu_baseline).  No reference implementation of these ops exists
(SURVEY.md section 0): parity is unpinned, these tests pin the oracle to the spec.
"""
import os

import numpy as np
import pytest
from scipy.signal import upfirdn

from util import bits_equal, frame_rel_err, rel_err, synth_clip

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_frame_count_bookkeeping(po):
    # n_frames = 1 + floor((n - n_fft)/hop), 0 if n < n_fft (docs/OPS.md "STFT")
    assert po.stft_n_frames(160000, 1024, 256) == 622          # BASELINE config 1
    assert po.stft_n_frames(1440000, 1024, 256) == 5622        # BASELINE config 2
    assert po.stft_n_frames(480000, 1024, 256) == 1872         # config 4 after 48k -> 16k
    assert po.stft_n_frames(1023, 1024, 256) == 0
    assert po.stft_n_frames(1024, 1024, 256) == 1
    assert po.stft_n_frames(1279, 1024, 256) == 1
    assert po.stft_n_frames(1280, 1024, 256) == 2
    assert po.stft_n_frames(0, 1024, 256) == 0


def test_resample_length_bookkeeping(po):
    assert po.resample_ratio(44100, 16000) == (160, 441)
    assert po.resample_ratio(48000, 16000) == (1, 3)
    assert po.resample_ratio(16000, 48000) == (3, 1)
    assert po.resample_out_len(441000, 160, 441) == 160000     # BASELINE config 3
    assert po.resample_out_len(1440000, 1, 3) == 480000
    assert po.resample_out_len(128, 1, 3) == 43                # one reference chunk: ceil(128/3)
    assert po.resample_out_len(1, 160, 441) == 1
    assert po.resample_out_len(0, 160, 441) == 0


def test_hann_is_periodic(po):
    w = po.hann_periodic(1024)
    assert w[0] == 0.0 and abs(w[512] - 1.0) < 1e-15
    assert np.allclose(w[1:], w[1:][::-1], atol=1e-15)         # w[n] == w[N-n]
    from scipy.signal import get_window
    assert np.allclose(w, get_window("hann", 1024, fftbins=True), atol=1e-15)


def test_stft_fft_vs_naive_dft_and_numpy(po):
    x = synth_clip(5, 1024 + 3 * 256, 1, 16000)
    P = po.stft_power(x)
    assert P.shape == (4, 513)
    w = po.hann_periodic(1024)
    for f in range(4):
        fr = x[f * 256:f * 256 + 1024]
        assert rel_err(P[f], po.dft_power_naive(fr, 1024)) < 1e-12
        assert rel_err(P[f], np.abs(np.fft.rfft(fr.astype(np.float64) * w)) ** 2) < 1e-13


@pytest.mark.parametrize("n_fft,hop", [(400, 160), (480, 240), (100, 64), (1000, 512), (33, 64), (600, 1024)])
def test_stft_of_frame_lengths_that_are_not_powers_of_two(po, n_fft, hop):
    """`afo_stft_power_f64` takes the DFT sum itself for these (the radix-2 FFT does not apply): pinned against numpy's
    mixed-radix rfft and the one-frame naive DFT; the whole chain (`po.features`) follows."""
    x = synth_clip(8, n_fft + 5 * hop + 7, 1, 16000)
    P = po.stft_power(x, n_fft, hop)
    assert P.shape == (6, n_fft // 2 + 1)
    w = po.hann_periodic(n_fft)
    for f in range(6):
        fr = x[f * hop:f * hop + n_fft]
        assert rel_err(P[f], np.abs(np.fft.rfft(fr.astype(np.float64) * w)) ** 2) < 1e-12
        assert rel_err(P[f], po.dft_power_naive(fr, n_fft)) < 1e-12
    W = po.mel_weights(16000.0, n_fft, 40)
    assert rel_err(po.features(x, 1, 16000.0, n_fft=n_fft, hop=hop, n_mels=40), P @ W.T) < 1e-12


def test_mel_weights_properties(po):
    W = po.mel_weights(48000.0)
    assert W.shape == (80, 513)
    assert W.min() >= 0.0 and W.max() <= 1.0
    # HTK scale round trip and the textbook value at 1 kHz
    assert abs(po.hz_to_mel(1000.0) - 999.9855) < 1e-3
    assert abs(po.mel_to_hz(po.hz_to_mel(4321.0)) - 4321.0) < 1e-9
    # between the first and the last centre the triangles sum to one
    edges = [po.mel_to_hz(po.hz_to_mel(24000.0) * i / 81) for i in range(82)]
    f = np.arange(513) * 48000.0 / 1024
    inside = (f >= edges[1]) & (f <= edges[80])
    assert np.allclose(W.sum(0)[inside], 1.0, atol=1e-12)
    # each filter's support is the open interval between its outer edges
    for m in (0, 17, 79):
        nz = np.nonzero(W[m])[0]
        if nz.size:
            assert f[nz[0]] > edges[m] and f[nz[-1]] < edges[m + 2]


def test_features_golden_48k(po):
    g = np.load(os.path.join(GOLD, "features_48k_stereo.npz"))
    mel = po.features(g["pcm"], 2, 48000.0)
    assert mel.shape == g["mel"].shape == (20, 80)
    assert bits_equal(mel, g["mel"])                        # the oracle is deterministic


def test_features_golden_16k_config1(po):
    g = np.load(os.path.join(GOLD, "features_16k_mono.npz"))
    assert po.stft_n_frames(int(g["n_samples_full"])) == int(g["n_frames_full"]) == 622
    mel = po.features(g["pcm"], 1, 16000.0)
    assert bits_equal(mel, g["mel"])


def test_float32_port_matches_float64_oracle(po):
    # the timed cpu_baseline ("port") computes the same thing within f32 rounding
    ch, n, sr = 2, 24000, 48000
    clips = [synth_clip(i, n, ch, sr) for i in range(3)]
    got = po.features_f32_batch(np.concatenate(clips), 3, n, ch, float(sr), threads=2)
    for i in range(3):
        ref = po.features(clips[i], ch, float(sr))
        assert frame_rel_err(got[i], ref) < 1e-5


@pytest.mark.parametrize("name", ["resample_44k1_16k", "resample_48k_16k", "resample_16k_48k"])
def test_resample_golden_and_scipy(po, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    x, L, M, T = g["x"], int(g["L"]), int(g["M"]), int(g["T"])
    y = po.resample(x, int(g["in_hz"]), int(g["out_hz"]))
    assert y.size == po.resample_out_len(x.size, L, M) == g["y"].size
    assert bits_equal(y, g["y"])
    h = po.resample_design(L, M, T)
    c = (T // 2) * L
    j0 = -(-c // M)
    yf = upfirdn(np.concatenate([np.zeros(j0 * M - c), h]), x.astype(np.float64), up=L, down=M)
    yf = np.concatenate([yf, np.zeros(max(0, j0 + y.size - yf.size))])
    assert np.max(np.abs(y - yf[j0:j0 + y.size])) < 1e-12


def test_resample_design_properties(po):
    L, M, T = 160, 441, 64
    h = po.resample_design(L, M, T)
    for r in (0, 1, 79, 159):                                   # unit DC gain per phase
        assert abs(h[r::L].sum() - 1.0) < 1e-12
    c = L * T // 2
    assert np.allclose(h[c + 1:], h[1:c][::-1], atol=1e-15)     # symmetric about L*T/2
    # a constant stays constant away from the edges, a sine keeps its amplitude
    y = po.resample(np.ones(4410, np.float32), 44100, 16000)
    assert np.allclose(y[40:-40], 1.0, atol=1e-12)
    t = np.arange(44100) / 44100.0
    y = po.resample(np.sin(2 * np.pi * 1000.0 * t).astype(np.float32), 44100, 16000)
    tt = np.arange(y.size) / 16000.0
    assert np.max(np.abs(y[100:-100] - np.sin(2 * np.pi * 1000.0 * tt)[100:-100])) < 2e-4


def test_resample_streaming_equals_one_shot(po):
    # outputs are a pure function of the absolute output index: any window of them matches
    x = synth_clip(9, 3000, 1, 48000)
    full = po.resample(x, 48000, 16000)
    part = po.resample(x, 48000, 16000, m0=137, n_out=400)
    assert bits_equal(part, full[137:537])


def test_fir_golden_and_numpy(po):
    g = np.load(os.path.join(GOLD, "fir_129.npz"))
    y = po.fir(g["x"], g["taps"])
    assert bits_equal(y, g["y"])
    ref = np.convolve(g["x"].astype(np.float64), g["taps"].astype(np.float64))[: g["x"].size]
    assert np.max(np.abs(y - ref)) < 1e-12
    assert rel_err(po.fir_f32(g["x"], g["taps"]), y) < 1e-5


def test_empty_and_short_inputs(po):
    assert po.stft_power(np.zeros(100, np.float32)).shape == (0, 513)
    assert po.features(np.zeros(2000, np.float32), 2, 48000.0).shape == (0, 80)
    assert po.resample(np.zeros(0, np.float32), 48000, 16000).size == 0
    assert po.fir(np.zeros(0, np.float32), [1.0]).size == 0


def test_level_meter_oracle(po):
    r, p, d = po.level_meter(np.full(800, 0.5, np.float32))
    assert abs(r - 0.5) < 1e-12 and p == 0.5 and abs(d - 20 * np.log10(0.5)) < 1e-12
    r, p, d = po.level_meter(np.zeros(10, np.float32))
    assert r == 0.0 and p == 0.0 and d == float("-inf")
    x = np.array([0.1, -0.9, 0.3], np.float32)
    r, p, d = po.level_meter(x)
    assert abs(r - np.sqrt(np.mean(x.astype(np.float64) ** 2))) < 1e-15 and p == float(np.float32(0.9))


def test_slaney_mel_weights_against_numpy(po):
    # independent restatement of docs/OPS.md "Mel filterbank" (Slaney scale + area norm)
    def hz2mel(f):
        f = np.asarray(f, float)
        return np.where(f >= 1000, 15 + np.log(np.maximum(f, 1e-30) / 1000) / (np.log(6.4) / 27), f / (200 / 3))

    def mel2hz(m):
        m = np.asarray(m, float)
        return np.where(m >= 15, 1000 * np.exp((np.log(6.4) / 27) * (m - 15)), (200 / 3) * m)

    e = mel2hz(np.linspace(hz2mel(0.0), hz2mel(8000.0), 82))
    f = np.arange(513) * 16000 / 1024
    Wn = np.zeros((80, 513))
    for m in range(80):
        up = (f - e[m]) / (e[m + 1] - e[m])
        dn = (e[m + 2] - f) / (e[m + 2] - e[m + 1])
        Wn[m] = np.maximum(0, np.minimum(up, dn)) * 2 / (e[m + 2] - e[m])
    assert np.max(np.abs(po.mel_weights(16000.0, scale=1, norm=1) - Wn)) < 1e-15
    # HTK default unchanged by the new entry point
    assert bits_equal(po.mel_weights(16000.0), po.mel_weights(16000.0, scale=0, norm=0))


def test_cubic_mode_formula_against_numpy_polyfit(po):
    """The cubic mode's closed formula == the interpolating cubic through the four neighbours (numpy polyfit)."""
    rng = np.random.default_rng(21)
    x = rng.uniform(-1, 1, 200).astype(np.float32)
    for fi, fo in [(48000, 16000), (16000, 48000), (44100, 16000)]:
        y = po.resample_cubic(x, fi, fo)
        L, M = po.resample_ratio(fi, fo)
        assert y.size == po.resample_out_len(x.size, L, M)
        xp = np.concatenate([[0.0], x.astype(np.float64), [0.0, 0.0, 0.0]])      # zero extension
        for m in range(0, y.size, 7):
            pos = m * fi / fo
            i = int(pos)
            c = np.polyfit([-1.0, 0.0, 1.0, 2.0], xp[i:i + 4], 3)
            assert abs(np.polyval(c, pos - i) - y[m]) < 1e-12
