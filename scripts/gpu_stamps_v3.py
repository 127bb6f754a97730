"""In-kernel phase stamps of features_v3 at BASELINE config 2: waves 0..3 and (second pass) waves 4..7."""
import ctypes as C, os, subprocess, sys
import numpy as np
if len(sys.argv) < 2:
    for w0 in ("0", "4"):
        subprocess.run([sys.executable, __file__, w0], env=dict(os.environ, AF_V3_STAMP_W0=w0), check=True)
    sys.exit(0)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import audio_flow_rs_amd as af
lib = af.library()
n_clips, nfr = 256, 1440000
dev = torch.device("cuda:0")
pcm = (torch.rand(n_clips, nfr, 2, device=dev) * 2 - 1) * 0.3
stream = torch.cuda.current_stream().cuda_stream
lib.af_features_debug_stamps.restype = C.c_int32; lib.af_features_debug_stamps.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
pipe = af.FeaturePipeline(sample_rate=48000, channels=2)
assert lib.af_features_set_option(pipe._h, b"kernel", 3) == 0
tot = pipe.set_batch_uniform(n_clips, nfr)
mel = torch.empty(tot, 80, device=dev)
for _ in range(20): pipe.run_device(pcm.data_ptr(), mel.data_ptr(), stream)
torch.cuda.synchronize()
assert lib.af_features_debug_stamps(pipe._h, 1, None) == 0
for _ in range(3): pipe.run_device(pcm.data_ptr(), mel.data_ptr(), stream)
torch.cuda.synchronize()
out = np.zeros(32, np.uint64)
assert lib.af_features_debug_stamps(pipe._h, 0, out.ctypes.data) == 0
per = out.reshape(4, 8).astype(np.float64) / (3 * tot / 16)
names = ["desc", "clip start", "mel (previous unit)", "stage 1", "barrier waits", "tile update", "stage 2", "next piece + issue"]
w0 = int(sys.argv[1])
print(f"cycles per unit, waves {w0}..{w0 + 3}:")
for i, nm in enumerate(names): print(f"  {nm:26s}", " ".join(f"{per[w, i]:8.0f}" for w in range(4)))
print("  total                     ", " ".join(f"{per[w].sum():8.0f}" for w in range(4)))
