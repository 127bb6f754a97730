"""The hop-256 kernel's mel phase on 4 x 4 x 1 blocks (MEL4) against the 16 x 16 x 4 plan (af_features_set_option "mel4" 0) and the
oracle; timing of both on BASELINE config 2 in one process.   python3 scripts/gpu_check_mel4.py [--time]"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import audio_flow_rs_amd as af
from oracle import pyoracle as po
from util import synth_clip, frame_rel_err
lib = af.library(); bad = 0
for kw in [dict(), dict(n_mels=40), dict(n_mels=128, sample_rate=44100), dict(n_mels=64, sample_rate=16000), dict(log_mel=True),
           dict(mel_scale=1, mel_norm=1), dict(n_fft=512), dict(channel_mode=1), dict(sample_format=1), dict(n_mels=24, sample_rate=8000)]:
    sr = kw.get("sample_rate", 48000); ch = 2
    lens = [20000, 1024 + 256 * 16 - 8, 1024, 900, 50000]
    clips = [synth_clip(300 + i, n, ch, sr) for i, n in enumerate(lens)]
    if kw.get("sample_format"): clips = [np.round(c * 32767.0).astype(np.int16) for c in clips]
    x = np.concatenate([c.reshape(-1) for c in clips]); off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    out = []
    for m4 in (1, 0):
        p = af.FeaturePipeline(channels=ch, **kw)
        assert lib.af_features_set_option(p._h, b"mel4", m4) == 0
        p.set_batch(off); out.append(p.run(x))
    d = frame_rel_err(out[0], np.asarray(out[1], np.float64)) if not kw.get("log_mel") else float(np.abs(out[0] - out[1]).max())
    ref_err = None
    if not kw.get("log_mel") and not kw.get("channel_mode"):
        W = po.mel_weights(float(sr), kw.get("n_fft", 1024), kw.get("n_mels", 80), scale=kw.get("mel_scale", 0), norm=kw.get("mel_norm", 0))
        row, ref_err = 0, 0.0
        for c, n in zip(clips, lens):
            cf = c.astype(np.float32) / np.float32(32768.0) if kw.get("sample_format") else c
            ref = po.mel_apply(po.stft_power(po.to_mono(cf, ch), kw.get("n_fft", 1024), 256), W)
            if ref.shape[0]: ref_err = max(ref_err, frame_rel_err(out[0][row:row + ref.shape[0]], ref))
            row += ref.shape[0]
    if "--map" in sys.argv and d > 2e-6:
        a, b = np.asarray(out[0], np.float64), np.asarray(out[1], np.float64)
        nm = a.shape[1]; e = np.abs(a - b) / (np.abs(b).max() + 1e-30)
        print("  per quad :", " ".join(f"{e[:, 4 * q:4 * q + 4].max():.1e}" for q in range(nm // 4)))
        print("  per frame (first clip, mod 16):", " ".join(f"{e[f:76:16].max():.1e}" for f in range(16)))
        print("  per mel%4:", " ".join(f"{e[:, i::4].max():.1e}" for i in range(4)))
        r = a[:8, :8] / (b[:8, :8] + 1e-30); print("  ratio[:8,:8]\n", np.array2string(r, precision=3))
    print(kw, "mel4 vs 16x16x4:", f"{d:.3e}", "vs oracle:", ref_err if ref_err is None else f"{ref_err:.3e}", "finite", bool(np.isfinite(out[0]).all()))
    bad += (d > 2e-6) or (ref_err is not None and ref_err > 1e-5)
if "--time" in sys.argv:
    import torch
    dev = torch.device("cuda:0")
    x = (torch.rand(256, 1440000, 2, device=dev) * 2 - 1) * 0.3
    s = torch.cuda.current_stream().cuda_stream
    pipes, res = {}, {0: [], 1: []}
    for m4 in (0, 1):                                    # (--pool: the same handle twice, work pool off / on)
        p = af.FeaturePipeline(); lib.af_features_set_option(p._h, b"pool" if "--pool" in sys.argv else b"mel4", m4)
        tot = p.set_batch_uniform(256, 1440000); pipes[m4] = p
    mel = torch.empty(tot, 80, device=dev)
    t0 = time.time()
    while time.time() - t0 < 1.0: pipes[0].run_device(x.data_ptr(), mel.data_ptr(), s); torch.cuda.synchronize()
    for rep in range(8):
        for m4 in (0, 1):
            p = pipes[m4]
            for _ in range(5): p.run_device(x.data_ptr(), mel.data_ptr(), s)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): p.run_device(x.data_ptr(), mel.data_ptr(), s)
            e1.record(); torch.cuda.synchronize()
            res[m4].append(e0.elapsed_time(e1) / 50)
    for m4 in (0, 1):
        r = sorted(res[m4]); print(f"config 2, {'pool' if '--pool' in sys.argv else 'mel4'}={m4}: min {r[0]:.4f}  median {r[len(r) // 2]:.4f}  max {r[-1]:.4f} ms  (8 x 50 launches, alternating)")
print("FAILED" if bad else "ok"); sys.exit(1 if bad else 0)
