"""Resampler timings for ratios with many block matrices (48 k <-> 44.1 k ...), 1024 x 30 s mono, device-resident:
   python3 scripts/gpu_bench_resample_ratios.py [valu]   ('valu': force the thread-per-output kernels)
   AF_BENCH_LIB=build/ablate/libaf_X.so: time another build of the library (scripts/ablate/build.sh); AF_BENCH_ONLY=48000_44100"""
import ctypes as C, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audio_flow_rs_amd as af
lib = af._ffi.load(os.environ.get("AF_BENCH_LIB")) if os.environ.get("AF_BENCH_LIB") else af.library(); dev = torch.device("cuda:0"); P = C.c_void_p
if len(sys.argv) > 1 and sys.argv[1] == "valu": lib.af_debug_set(b"banded.off", 1)
def timeit(fn, k=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15: fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
out = {}
only = os.environ.get("AF_BENCH_ONLY")
for fi, fo, nc, secs in [(48000, 44100, 1024, 30), (44100, 48000, 1024, 30), (48000, 22050, 1024, 30), (16000, 44100, 1024, 30),
                         (44100, 16000, 4096, 10), (48000, 16000, 1024, 30)]:
    if only and only != f"{fi}_{fo}": continue
    nin = fi * secs
    nout = lib.af_resample_out_len(fi, fo, nin)
    x = torch.rand(nc, nin, device=dev) - 0.5; y = torch.empty(nc, nout, device=dev)
    ms = timeit(lambda: lib.af_resample_batch_device(fi, fo, P(x.data_ptr()), nc, nin, P(y.data_ptr()), None))
    b = (x.numel() + y.numel()) * 4
    out[f"resample_{fi}_{fo}"] = dict(clips=nc, seconds=secs, ms=round(ms, 4), alg_GBps=round(b / ms / 1e6, 1),
                                      frac_hbm=round(b / ms / 1e6 / 8000, 4), tflops_useful=round(2 * 64 * y.numel() / ms / 1e9, 1))
    del x, y
print(json.dumps(out, indent=1))
