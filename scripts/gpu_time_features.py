"""Quick timing of the fused kernel at BASELINE config 2 using torch device memory."""
import ctypes as C, importlib.util, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("ffi", os.path.join(ROOT, "audio-flow-rs_amd", "_ffi.py"))
ffi = importlib.util.module_from_spec(spec); spec.loader.exec_module(ffi)
lib = C.CDLL(os.environ.get("AF_LIB", ffi.LIB_PATH))
for name in ["af_debug_set", "af_last_error", "af_features_default_config", "af_features_create", "af_features_destroy",
             "af_features_n_frames", "af_features_set_batch", "af_features_set_batch_uniform",
             "af_features_run", "af_features_run_device"]:
    r, a = ffi.SIGNATURES[name]; fn = getattr(lib, name); fn.restype = r; fn.argtypes = a
def check(st):
    if st != 0: raise RuntimeError(lib.af_last_error().decode())
n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 256
GENERAL = bool(os.environ.get("AF_FEATURES_GENERAL"))          # script-side switch only
lib.af_debug_set(b"features.kernel", 1 if GENERAL else 0)
if os.environ.get("AF_FEATURES_WG_PER_CU"): lib.af_debug_set(b"features.wg_per_cu", int(os.environ["AF_FEATURES_WG_PER_CU"]))
n = 1440000
dev = torch.device("cuda:0")
pcm = (torch.rand(n_clips, n, 2, device=dev) * 2 - 1) * 0.3
cfg = ffi.FeaturesConfig(); lib.af_features_default_config(C.byref(cfg))
h = C.c_void_p(); check(lib.af_features_create(C.byref(cfg), C.byref(h)))
tot = C.c_size_t(); check(lib.af_features_set_batch_uniform(h, n_clips, n, C.byref(tot)))
mel = torch.empty(tot.value, 80, device=dev)
stream = torch.cuda.current_stream().cuda_stream
for _ in range(30): check(lib.af_features_run_device(h, pcm.data_ptr(), mel.data_ptr(), stream))   # past the clock ramp
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
K = 30
e0.record()
for _ in range(K): check(lib.af_features_run_device(h, pcm.data_ptr(), mel.data_ptr(), stream))
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / K
bytes_alg = n_clips * n * 8 + tot.value * 320
print(f"clips={n_clips} frames={tot.value} ms={ms:.3f} frames/s={tot.value/ms*1e3:.3e} RTF={n_clips*30/ms*1e3:.3e} alg GB/s={bytes_alg/ms/1e6:.1f} frac8TB={bytes_alg/ms/1e6/8000:.3f}")
print("mel finite:", bool(torch.isfinite(mel).all()), float(mel.abs().mean()))
# ---- phase stamps (diagnostic)
import numpy as np
lib.af_features_debug_stamps.restype = C.c_int32; lib.af_features_debug_stamps.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
check(lib.af_features_debug_stamps(h, 1, None))
check(lib.af_features_run_device(h, pcm.data_ptr(), mel.data_ptr(), stream)); torch.cuda.synchronize()
out = np.zeros(32, np.uint64)
check(lib.af_features_debug_stamps(h, 0, out.ctypes.data))
units = tot.value / 16
names = ["desc load", "fresh+B1", "pf issue+part1", "wf issue+part2", "B2 wait", "mel mfma+stores", "tile write+B reads", "wf landed"] if not GENERAL else ["B1 wait", "prefetch issue", "FFT phase", "B2 wait", "tile write", "mel operands", "mel mfma+store", "loop tail"]
per = out.reshape(4, 8).astype(np.float64) / units
print("cycles per unit (iteration), per wave:")
for i, nm in enumerate(names): print(f"  {nm:15s}", " ".join(f"{per[w, i]:9.0f}" for w in range(4)))
print("  total          ", " ".join(f"{per[w].sum():9.0f}" for w in range(4)))
