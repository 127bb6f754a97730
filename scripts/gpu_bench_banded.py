"""A/B timings of the banded MFMA kernel (config 3 resample, 48k->16k, FIR 129)."""
import ctypes as C, os, sys, json, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audio_flow_rs_amd as af
lib = af.library(); dev = torch.device("cuda:0"); P = C.c_void_p
def timeit(fn, k=5, w=2):
    # the chip ramps its clocks for ~25 launches / ~50 ms after idle: warm up past that, whatever w says
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15: fn(); torch.cuda.synchronize()
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
which = sys.argv[1:] or ["r441", "r48", "fir"]
out = {}
if "r441" in which:
    nc, nin = 4096, 441000
    nout = lib.af_resample_out_len(44100, 16000, nin)
    x = torch.rand(nc, nin, device=dev) - 0.5; y = torch.empty(nc, nout, device=dev)
    ms = timeit(lambda: lib.af_resample_batch_device(44100, 16000, P(x.data_ptr()), nc, nin, P(y.data_ptr()), None))
    b = (x.numel() + y.numel()) * 4
    out["resample_44k1_16k_cfg3"] = dict(ms=round(ms, 3), frac_hbm=round(b / ms / 1e6 / 8000, 3))
    del x, y
nc, nin = 1024, 1440000
nout = lib.af_resample_out_len(48000, 16000, nin)
y = torch.empty(nc, nout, device=dev)
if "r48" in which:
    x = torch.rand(nc, nin, device=dev) - 0.5
    ms = timeit(lambda: lib.af_resample_batch_device(48000, 16000, P(x.data_ptr()), nc, nin, P(y.data_ptr()), None))
    b = (x.numel() + y.numel()) * 4
    out["resample_48k_16k_cfg4"] = dict(ms=round(ms, 3), frac_hbm=round(b / ms / 1e6 / 8000, 3))
    del x
if "fir" in which:
    y.uniform_(-0.5, 0.5)
    taps = torch.rand(129, device=dev) / 129; z = torch.empty_like(y)
    ms = timeit(lambda: lib.af_fir_batch_device(P(taps.data_ptr()), 129, P(y.data_ptr()), nc, nout, P(z.data_ptr()), None))
    b = 2 * y.numel() * 4
    out["fir129_cfg4"] = dict(ms=round(ms, 3), frac_hbm=round(b / ms / 1e6 / 8000, 3), tflops=round(2 * 129 * y.numel() / ms / 1e9, 1))
print(os.environ.get("AF_BANDED_DBG", "0"), json.dumps(out))
