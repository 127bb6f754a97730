"""Stereo input through the resampler: downmix inside the resampler's loads (af_resample_batch_stereo_device) against
af_to_mono_device + af_resample_batch_device, 256 x 30 s stereo, device-resident:
   python3 scripts/gpu_bench_stereo_resample.py [in_hz out_hz ...]"""
import ctypes as C, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audio_flow_rs_amd as af
lib = af.library(); dev = torch.device("cuda:0"); P = C.c_void_p
def timeit(fn, k=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15: fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
args = [int(a) for a in sys.argv[1:]] or [44100, 16000, 44100, 48000, 48000, 16000]
out = {}
for fi, fo in zip(args[0::2], args[1::2]):
    nc, nin = 256, fi * 30
    nout = lib.af_resample_out_len(fi, fo, nin)
    x = torch.rand(nc, nin, 2, device=dev) - 0.5
    mono = torch.empty(nc, nin, device=dev); y = torch.empty(nc, nout, device=dev); y2 = torch.empty(nc, nout, device=dev)
    def two():
        assert lib.af_to_mono_device(P(x.data_ptr()), nc * nin * 2, 2, P(mono.data_ptr()), None) == 0
        assert lib.af_resample_batch_device(fi, fo, P(mono.data_ptr()), nc, nin, P(y.data_ptr()), None) == 0
    def one():
        assert lib.af_resample_batch_stereo_device(fi, fo, P(x.data_ptr()), nc, nin, P(y2.data_ptr()), None) == 0
    r = {"has_stereo_path": int(lib.af_resample_has_stereo_path(fi, fo)), "to_mono_plus_resample_ms": round(timeit(two), 4)}
    if r["has_stereo_path"]:
        r["stereo_resample_ms"] = round(timeit(one), 4)
        r["max_abs_diff"] = float((y - y2).abs().max())
        b = (x.numel() + y.numel()) * 4
        r["frac_hbm"] = round(b / r["stereo_resample_ms"] / 1e6 / 8000, 4)
    out[f"{fi}_{fo}"] = r
print(json.dumps(out, indent=1))
