"""Randomised configurations of the fused feature path against the float64 oracle (frame counts exact, rows within 1e-5 per frame):
frame lengths 32..1024 (the three register-FFT lengths and the matrix-product DFT), hops, mel counts, channels, formats, ragged batches.
   python3 scripts/gpu_fuzz_features.py [N=80] [seed=1]"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import audio_flow_rs_amd as af
from oracle import pyoracle as po
from util import synth_clip, frame_rel_err
N = int(sys.argv[1]) if len(sys.argv) > 1 else 80
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
worst, bad = 0.0, 0
for it in range(N):
    n_fft = int(rng.choice([1024, 512, 256, 400, 480, 320, 200, 100, 33, 64, 600, 1000, 777, 36]))
    hop = int(rng.choice([64, 80, 128, 160, 240, 256, 320, 512, 1024]))
    sr = int(rng.choice([8000, 16000, 22050, 44100, 48000]))
    ch = int(rng.choice([1, 2, 2, 3]))
    fmt = int(rng.choice([0, 0, 1])) if ch <= 2 else 0
    n_mels = int(rng.choice([80, 40, 64, 128, 23, 13]))
    sep = int(ch == 2 and rng.random() < 0.2)
    kw = dict(sample_rate=sr, channels=ch, n_fft=n_fft, hop=hop, n_mels=n_mels, sample_format=fmt, channel_mode=sep)
    try:
        pipe = af.FeaturePipeline(**kw)
    except af.AudioError as e:
        print(it, kw, "refused:", str(e)[:80]); continue
    lens = [int(rng.integers(1, n_fft + hop * 40)) for _ in range(int(rng.integers(1, 6)))]
    clips = [synth_clip(1000 + 10 * it + i, n, ch, sr) for i, n in enumerate(lens)]
    if fmt: clips = [np.round(c * 32767.0).astype(np.int16) for c in clips]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    total = pipe.set_batch(off)
    want = [(2 if sep else 1) * (1 + (n - n_fft) // hop if n >= n_fft else 0) for n in lens]
    assert total == sum(want), (kw, lens, total, want)
    got = pipe.run(np.concatenate(clips)) if total else np.zeros((0, n_mels), np.float32)
    W = po.mel_weights(float(sr), n_fft, n_mels)
    row, err = 0, 0.0
    for c, nrows in zip(clips, want):
        if nrows:
            x = c.astype(np.float32) / np.float32(32768.0) if fmt else c
            if sep:
                xs = x.reshape(-1, ch)
                ref = np.concatenate([po.mel_apply(po.stft_power(np.ascontiguousarray(xs[:, k]), n_fft, hop), W) for k in range(ch)])
            else:
                ref = po.mel_apply(po.stft_power(po.to_mono(x, ch), n_fft, hop), W)
            err = max(err, frame_rel_err(got[row:row + nrows], ref))
        row += nrows
    worst = max(worst, err)
    flag = "" if err < 1e-5 and np.isfinite(got).all() else "   <-- FAIL"
    bad += bool(flag)
    print(it, kw, "clips", lens, f"err {err:.2e}{flag}", flush=True)
print(f"{N} configurations, worst per-frame error {worst:.2e}, failures {bad}")
sys.exit(1 if bad else 0)
