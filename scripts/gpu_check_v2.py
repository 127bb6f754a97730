"""First-light check of the hop-256 kernel (features_v2.hip) on the GPU: aligned ragged batches in all
four PCM formats against the float64 oracle and against the general kernel (af_debug_set features.kernel = 1),
power / log-mel / separate-channel outputs, then config-2 timing of both kernels."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import audio_flow_rs_amd as af
if os.environ.get('AF_LIB'): af._ffi.LIB_PATH = os.environ['AF_LIB']
from oracle import pyoracle as po
from util import synth_clip, frame_rel_err

def run(clips, ch, fmt=0, general=False, **kw):
    af.library().af_debug_set(b"features.kernel", 1 if general else 0)
    pipe = af.FeaturePipeline(sample_rate=48000, channels=ch, hop=256, sample_format=fmt, **kw)
    offs = np.zeros(len(clips) + 1, np.int64)
    for i, c in enumerate(clips): offs[i + 1] = offs[i] + c.size // ch
    pipe.set_batch(offs)
    out = pipe.run(np.concatenate(clips))
    af.library().af_debug_set(b"features.kernel", 0)
    return pipe, out

worst = 0.0
lens = [0, 1016, 1024, 1024 + 256 * 15, 1024 + 256 * 16, 1024 + 256 * 16 - 8, 7776, 50000, 8, 1024 + 256 * 47, 123456]
for ch in (2, 1):
    for fmt in (0, 1):
        clips = [synth_clip(i, n, ch, 48000) for i, n in enumerate(lens)]
        if fmt: clips = [np.round(c * 32767.0).astype(np.int16) for c in clips]
        pipe, mel = run(clips, ch, fmt)
        _, melg = run(clips, ch, fmt, general=True)
        row = 0
        for n, x in zip(lens, clips):
            xf = x.astype(np.float32) / np.float32(32768.0) if fmt else x
            ref = po.features(xf, ch, 48000.0)
            nf = ref.shape[0]
            e = frame_rel_err(mel[row:row + nf], ref) if nf else 0.0
            eg = frame_rel_err(mel[row:row + nf], melg[row:row + nf]) if nf else 0.0
            worst = max(worst, e)
            if e > 1e-5:
                g = np.asarray(mel[row:row + nf], np.float64)
                fe = np.linalg.norm(g - ref, axis=1) / np.linalg.norm(ref, axis=1)
                bad = np.nonzero(fe > 1e-5)[0]
                print("   bad frames:", bad.tolist()[:64], "first bad values", g[bad[0]][:6], "ref", ref[bad[0]][:6], flush=True)
            print(f"ch={ch} fmt={fmt} n={n:7d} frames={nf:4d} err_vs_oracle={e:.2e} vs_general={eg:.2e}", flush=True)
            row += nf
        assert row == mel.shape[0] and np.all(np.isfinite(mel))
print("worst vs oracle", worst)
# other outputs
x = [synth_clip(40 + i, 20000 + 4096 * i, 2, 48000) for i in range(3)]
for kw in (dict(output=1), dict(log_mel=1), dict(channel_mode=1), dict(n_mels=40), dict(n_mels=128, sample_rate=44100)):
    try:
        kw2 = dict(kw); sr = kw2.pop("sample_rate", 48000)
        def mk(general):
            af.library().af_debug_set(b"features.kernel", 1 if general else 0)
            p = af.FeaturePipeline(sample_rate=sr, channels=2, hop=256, **kw2)
            offs = np.cumsum([0] + [c.size // 2 for c in x]).astype(np.int64)
            p.set_batch(offs); o = p.run(np.concatenate(x)); af.library().af_debug_set(b"features.kernel", 0); return o
        a, b = mk(False), mk(True)
        print(kw, a.shape, "max rel diff vs general", float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30)), flush=True)
    except Exception as ex:
        print(kw, "FAILED", ex, flush=True)
assert worst < 1e-5
