"""First-light check of the fused STFT+mel kernel against the float64 oracle (GPU box)."""
import ctypes as C, importlib.util, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po
spec = importlib.util.spec_from_file_location("ffi", os.path.join(ROOT, "audio-flow-rs_amd", "_ffi.py"))
ffi = importlib.util.module_from_spec(spec); spec.loader.exec_module(ffi)
lib = C.CDLL(ffi.LIB_PATH)
for name in ["af_last_error", "af_features_default_config", "af_features_create", "af_features_destroy",
             "af_features_n_frames", "af_features_set_batch", "af_features_set_batch_uniform",
             "af_features_run", "af_features_run_device"]:
    r, a = ffi.SIGNATURES[name]; fn = getattr(lib, name); fn.restype = r; fn.argtypes = a

def check(st):
    if st != 0:
        raise RuntimeError(lib.af_last_error().decode())

def run(pcm_list, channels, sr=48000, hop=256, n_mels=80, fmt=0):
    cfg = ffi.FeaturesConfig(); lib.af_features_default_config(C.byref(cfg))
    cfg.sample_rate = sr; cfg.channels = channels; cfg.hop = hop; cfg.n_mels = n_mels; cfg.sample_format = fmt
    h = C.c_void_p(); check(lib.af_features_create(C.byref(cfg), C.byref(h)))
    offs = np.zeros(len(pcm_list) + 1, np.int64)
    for i, x in enumerate(pcm_list): offs[i + 1] = offs[i] + x.size // channels
    tot = C.c_size_t(); check(lib.af_features_set_batch(h, offs.ctypes.data, len(pcm_list), C.byref(tot)))
    pcm = np.ascontiguousarray(np.concatenate(pcm_list)) if pcm_list else np.zeros(0, np.float32)
    mel = np.full((tot.value, n_mels), np.nan, np.float32)
    check(lib.af_features_run(h, pcm.ctypes.data, pcm.size, mel.ctypes.data, mel.size))
    lib.af_features_destroy(h)
    return mel, offs

rng = np.random.default_rng(1)
def synth(n, ch):
    t = np.arange(n) / 48000.0
    x = sum(a * np.sin(2 * np.pi * f * t + p) for a, f, p in [(0.3, 440, 0.1), (0.1, 3000, 1.0), (0.05, 12000, 2.0)])
    x = x[:, None] + 0.01 * rng.uniform(-1, 1, (n, ch))
    return np.clip(x, -1, 1).astype(np.float32).reshape(-1)

ok = True
for ch, lens, hop in [(2, [1024 + 256 * 15, 48000, 1023, 1024, 5000, 0, 30000], 256),
                      (1, [16000, 2048, 777], 256), (2, [20000, 4000], 128), (1, [20000], 512), (2, [9000], 320)]:
    clips = [synth(n, ch) for n in lens]
    mel, offs = run(clips, ch, hop=hop)
    row = 0
    for x in clips:
        ref = po.features(x, ch, 48000.0, hop=hop)
        got = mel[row:row + ref.shape[0]]; row += ref.shape[0]
        if ref.shape[0] == 0: continue
        err = np.linalg.norm(got - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-30)
        print(f"ch={ch} hop={hop} n={x.size // ch:6d} frames={ref.shape[0]:4d} max per-frame rel err {err.max():.3e}")
        ok &= bool(np.all(np.isfinite(got))) and err.max() < 1e-5
    assert row == mel.shape[0]
print("PASS" if ok else "FAIL")
sys.exit(0 if ok else 1)
