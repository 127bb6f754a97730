"""Generates the committed golden vectors under tests/golden/.

The reference (forfd8960/audio-flow-rs) contains no STFT / mel / FIR / polyphase
code and no test vectors for them ("parity unpinned"), so
these fixtures are produced by the float64 oracle (oracle/af_oracle.c) and are
accepted only after agreeing with an INDEPENDENT numpy/scipy formulation of
docs/OPS.md, computed right here.  Run from the repo root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
from scipy.signal import upfirdn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle as po   # noqa: E402
from util import synth_clip          # noqa: E402


def numpy_mel_weights(sr, n_fft, n_mels, fmin, fmax):
    hz2mel = lambda f: 2595.0 * np.log10(1.0 + f / 700.0)
    mel2hz = lambda m: 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    edges = mel2hz(np.linspace(hz2mel(fmin), hz2mel(fmax), n_mels + 2))
    f = np.arange(n_fft // 2 + 1) * sr / n_fft
    W = np.zeros((n_mels, f.size))
    for m in range(n_mels):
        up = (f - edges[m]) / (edges[m + 1] - edges[m])
        dn = (edges[m + 2] - f) / (edges[m + 2] - edges[m + 1])
        W[m] = np.maximum(0.0, np.minimum(up, dn))
    return W


def numpy_features(pcm, ch, sr, n_fft=1024, hop=256, n_mels=80):
    x = pcm.reshape(-1, ch)
    mono = x[:, 0].copy() if ch == 1 else ((x[:, 0] + x[:, 1]) / np.float32(2)).astype(np.float32)
    nf = 1 + (mono.size - n_fft) // hop
    w = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)
    frames = np.stack([mono[i * hop:i * hop + n_fft].astype(np.float64) * w for i in range(nf)])
    P = np.abs(np.fft.rfft(frames, axis=1)) ** 2
    return P @ numpy_mel_weights(sr, n_fft, n_mels, 0.0, sr / 2).T


def scipy_resample(x, h, L, M, T):
    c = (T // 2) * L
    j0 = -(-c // M)
    z = j0 * M - c
    yf = upfirdn(np.concatenate([np.zeros(z), h]), x.astype(np.float64), up=L, down=M)
    n_out = -(-x.size * L // M)
    yf = np.concatenate([yf, np.zeros(max(0, j0 + n_out - yf.size))])
    return yf[j0:j0 + n_out]


def main():
    # 1. fused feature path, stereo 48 kHz, 6000 frames -> 20 x 80
    pcm = synth_clip(0, 6000, 2, 48000)
    mel = po.features(pcm, 2, 48000.0)
    ref = numpy_features(pcm, 2, 48000)
    assert np.max(np.abs(mel - ref)) <= 1e-9 * np.max(np.abs(ref)), "oracle vs numpy (features)"
    np.savez_compressed(os.path.join(HERE, "features_48k_stereo.npz"), pcm=pcm, mel=mel,
                        sample_rate=48000, channels=2, n_fft=1024, hop=256, n_mels=80)

    # 1b. BASELINE config 1 shape: 10 s mono 16 kHz -> 622 frames (power kept as mel at 16 kHz)
    pcm16 = synth_clip(1, 160000, 1, 16000)
    mel16 = po.features(pcm16, 1, 16000.0)
    assert mel16.shape == (622, 80)
    ref16 = numpy_features(pcm16, 1, 16000)
    assert np.max(np.abs(mel16 - ref16)) <= 1e-9 * np.max(np.abs(ref16))
    # keep the fixture small: first 4000 samples (12 frames) + the frame count of the full clip
    np.savez_compressed(os.path.join(HERE, "features_16k_mono.npz"), pcm=pcm16[:4000],
                        mel=mel16[:12], n_frames_full=622, n_samples_full=160000,
                        sample_rate=16000, channels=1, n_fft=1024, hop=256, n_mels=80)

    # 2./3. polyphase resampler
    for name, fi, fo, n in [("resample_44k1_16k", 44100, 16000, 2205), ("resample_48k_16k", 48000, 16000, 1500),
                            ("resample_16k_48k", 16000, 48000, 400)]:
        x = synth_clip(2, n, 1, fi)
        L, M = po.resample_ratio(fi, fo)
        h = po.resample_design(L, M, 64)
        y = po.resample(x, fi, fo)
        ys = scipy_resample(x, h, L, M, 64)
        assert y.size == ys.size == -(-n * L // M)
        assert np.max(np.abs(y - ys)) < 1e-12, name
        np.savez_compressed(os.path.join(HERE, name + ".npz"), x=x, y=y, L=L, M=M, T=64,
                            in_hz=fi, out_hz=fo, h_first_phase=h[::L][:64])

    # 4. FIR, 129-tap Hamming low-pass (cutoff 7.2 kHz at 16 kHz: BASELINE config 4)
    K = 129
    n = np.arange(K) - (K - 1) / 2
    fc = 7200.0 / 16000.0
    taps = (2 * fc * np.sinc(2 * fc * n) * np.hamming(K)).astype(np.float32)
    x = synth_clip(3, 3000, 1, 16000)
    y = po.fir(x, taps)
    yn = np.convolve(x.astype(np.float64), taps.astype(np.float64))[:x.size]
    assert np.max(np.abs(y - yn)) < 1e-12
    np.savez_compressed(os.path.join(HERE, "fir_129.npz"), x=x, taps=taps, y=y)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
