"""Runs the fused feature kernel a few times at BASELINE config 2 (for rocprofv3 passes)."""
import ctypes as C, importlib.util, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("ffi", os.path.join(ROOT, "audio-flow-rs_amd", "_ffi.py"))
ffi = importlib.util.module_from_spec(spec); spec.loader.exec_module(ffi)
lib = ffi.load(os.environ.get("AF_LIB"))
n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = 1440000
dev = torch.device("cuda:0")
pcm = (torch.rand(n_clips, n, 2, device=dev) * 2 - 1) * 0.3
cfg = ffi.FeaturesConfig(); lib.af_features_default_config(C.byref(cfg))
h = C.c_void_p(); assert lib.af_features_create(C.byref(cfg), C.byref(h)) == 0
if os.environ.get("AF_KERNEL"):        # 1: general kernel, 2: features_v2, 3: features_v3 -- asked for by THIS handle (the tables of the
    assert lib.af_features_set_option(h, b"kernel", int(os.environ["AF_KERNEL"])) == 0   # matrix-core kernel are built on request)
tot = C.c_size_t(); assert lib.af_features_set_batch_uniform(h, n_clips, n, C.byref(tot)) == 0
mel = torch.empty(tot.value, 80, device=dev)
s = torch.cuda.current_stream().cuda_stream
for _ in range(reps): assert lib.af_features_run_device(h, pcm.data_ptr(), mel.data_ptr(), s) == 0
torch.cuda.synchronize()
print("ok", tot.value)
