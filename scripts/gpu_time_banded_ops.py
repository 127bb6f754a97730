"""Times the banded-MFMA kernels (config 3 / config 4 stages, chain) for a library variant (AF_LIB)."""
import ctypes as C, importlib.util, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("ffi", os.path.join(ROOT, "audio-flow-rs_amd", "_ffi.py"))
ffi = importlib.util.module_from_spec(spec); spec.loader.exec_module(ffi)
lib = ffi.load(os.environ.get("AF_LIB"))
dev = torch.device("cuda:0"); P = C.c_void_p
def timeit(fn, k=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2: fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
nc, nin = 4096, 441000
nout = lib.af_resample_out_len(44100, 16000, nin)
x = torch.rand(nc, nin, device=dev) - 0.5; y = torch.empty(nc, nout, device=dev)
print(f"cfg3 44.1->16k: {timeit(lambda: lib.af_resample_batch_device(44100, 16000, P(x.data_ptr()), nc, nin, P(y.data_ptr()), None), 5):.4f} ms")
del x, y
nc, nin = 1024, 1440000
nout = lib.af_resample_out_len(48000, 16000, nin)
x = torch.rand(nc, nin, device=dev) - 0.5; y = torch.empty(nc, nout, device=dev); z = torch.empty_like(y)
print(f"48->16k mono: {timeit(lambda: lib.af_resample_batch_device(48000, 16000, P(x.data_ptr()), nc, nin, P(y.data_ptr()), None), 5):.4f} ms")
taps = torch.rand(129, device=dev) / 129
print(f"fir129: {timeit(lambda: lib.af_fir_batch_device(P(taps.data_ptr()), 129, P(y.data_ptr()), nc, nout, P(z.data_ptr()), None), 5):.4f} ms")
del x, y, z
n_clips, n = 256, 1440000
pcm = torch.rand(n_clips, n, 2, device=dev) - 0.5
k = np.arange(129) - 64
th = (0.9 * np.sinc(0.9 * k) * np.hamming(129)).astype(np.float32)
cfg = ffi.ChainConfig(); lib.af_chain_default_config(C.byref(cfg)); cfg.fir_taps = th.ctypes.data; cfg.n_fir_taps = 129
h = C.c_void_p(); assert lib.af_chain_create(C.byref(cfg), C.byref(h)) == 0
tot = C.c_size_t(); assert lib.af_chain_set_batch_uniform(h, n_clips, n, C.byref(tot)) == 0
mel = torch.empty(tot.value, 80, device=dev); s = torch.cuda.current_stream().cuda_stream
print(f"chain: {timeit(lambda: lib.af_chain_run_device(h, pcm.data_ptr(), mel.data_ptr(), s), 20):.4f} ms")
