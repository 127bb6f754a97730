import sys, os, ctypes as C, torch
sys.path.insert(0, "/root/repo")
import audio_flow_rs_amd as af
dev = torch.device("cuda:0")
tot = 1439232
pw = torch.rand(tot, 513, device=dev); ml = torch.empty(tot, 80, device=dev)
fb = af.MelFilterbank(sample_rate=48000)
st = torch.cuda.current_stream().cuda_stream
for _ in range(2): fb.apply_device(pw.data_ptr(), tot, ml.data_ptr(), st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): fb.apply_device(pw.data_ptr(), tot, ml.data_ptr(), st)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print("mel_apply ms", ms, "frac", (pw.numel() + ml.numel()) * 4 / ms / 1e6 / 8000)
