"""to_mono / quantise timing only (streaming-kernel tuning)."""
import ctypes as C, os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audio_flow_rs_amd as af
lib = af.library(); dev = torch.device("cuda:0"); P = C.c_void_p
def timeit(fn, k=10, w=2):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
s = torch.rand(256 * 1440000, 2, device=dev); m = torch.empty(256 * 1440000, device=dev)
ms = timeit(lambda: lib.af_to_mono_device(P(s.data_ptr()), s.numel(), 2, P(m.data_ptr()), None))
print("to_mono", round(ms, 3), round(s.numel() * 6 / ms / 1e6 / 8000, 3))
q = torch.empty(m.numel(), dtype=torch.int16, device=dev)
ms = timeit(lambda: lib.af_quantize_pcm16_device(P(m.data_ptr()), m.numel(), P(q.data_ptr()), None))
print("quantize", round(ms, 3), round(m.numel() * 6 / ms / 1e6 / 8000, 3))
t = torch.empty_like(m)
ms = timeit(lambda: t.copy_(m))
print("torch copy", round(ms, 3), round(m.numel() * 8 / ms / 1e6 / 8000, 3))
