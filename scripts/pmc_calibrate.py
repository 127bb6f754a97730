"""Known-byte-count streaming kernel (af_to_mono_device on 1 GiB of stereo f32) used to
calibrate FETCH_SIZE / WRITE_SIZE on gfx950 for 16-byte-per-lane reads and 8-byte stores
(MI355X_MICROARCH.md, HBM section).  Run under rocprofv3 --pmc."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audio_flow_rs_amd as af
n_frames = 128 * 1024 * 1024            # 1 GiB of stereo f32 in, 0.5 GiB out
x = torch.rand(n_frames, 2, device="cuda")
y = torch.empty(n_frames, device="cuda")
lib = af.library()
for _ in range(3):
    assert lib.af_to_mono_device(C.c_void_p(x.data_ptr()), x.numel(), 2, C.c_void_p(y.data_ptr()), None) == 0
torch.cuda.synchronize()
print("bytes_read", x.numel() * 4, "bytes_written", y.numel() * 4)
