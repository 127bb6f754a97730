"""Times the streaming kernels alone (VAD energy, level meter, quantise, base64, to_mono) on 1024 x 480000 samples.
AF_LIB selects a library variant.  usage: gpu_time_stream_ops.py [op ...]"""
import ctypes as C, importlib.util, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("ffi", os.path.join(ROOT, "audio-flow-rs_amd", "_ffi.py"))
ffi = importlib.util.module_from_spec(spec); spec.loader.exec_module(ffi)
lib = ffi.load(os.environ.get("AF_LIB"))
dev = torch.device("cuda:0"); P = C.c_void_p
ops = sys.argv[1:] or ["vad", "level", "quant", "b64", "mono"]
def timeit(fn, k=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15: fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
y = torch.rand(1024 * 480000, device=dev) - 0.5
n = y.numel()
if "vad" in ops:
    en = torch.empty(n // 320, device=dev)
    ms = timeit(lambda: lib.af_vad_energy_device(P(y.data_ptr()), n // 320, 320, P(en.data_ptr()), None))
    print(f"vad_energy_320 {ms:.4f} ms {n * 4 / ms / 1e6:.0f} GB/s")
    en2 = torch.empty(n // 512, device=dev)
    ms = timeit(lambda: lib.af_vad_energy_device(P(y.data_ptr()), n // 512, 512, P(en2.data_ptr()), None))
    print(f"vad_energy_512 {ms:.4f} ms {n * 4 / ms / 1e6:.0f} GB/s")
if "level" in ops:
    nw = n // 800
    lv = [torch.empty(nw, device=dev) for _ in range(3)]
    ms = timeit(lambda: lib.af_level_meter_device(P(y.data_ptr()), nw, 800, P(lv[0].data_ptr()), P(lv[1].data_ptr()), P(lv[2].data_ptr()), None))
    print(f"level_meter_800 {ms:.4f} ms {nw * 800 * 4 / ms / 1e6:.0f} GB/s")
if "quant" in ops:
    q = torch.empty(n, dtype=torch.int16, device=dev)
    ms = timeit(lambda: lib.af_quantize_pcm16_device(P(y.data_ptr()), n, P(q.data_ptr()), None))
    print(f"quantize {ms:.4f} ms {n * 6 / ms / 1e6:.0f} GB/s")
if "b64" in ops:
    b64 = torch.empty(lib.af_base64_len(2 * n) + 16, dtype=torch.uint8, device=dev)
    ms = timeit(lambda: lib.af_pcm16_base64_device(P(y.data_ptr()), n, P(b64.data_ptr()), None))
    print(f"pcm16_base64 {ms:.4f} ms {(n * 4 + b64.numel()) / ms / 1e6:.0f} GB/s")
if "mono" in ops:
    m = torch.empty(n // 2, device=dev)
    ms = timeit(lambda: lib.af_to_mono_device(P(y.data_ptr()), n, 2, P(m.data_ptr()), None))
    print(f"to_mono {ms:.4f} ms {n * 6 / ms / 1e6:.0f} GB/s")
