"""Runs BASELINE config 4's chain (stereo 48 kHz -> 16 kHz -> FIR 129 -> 80 mel) a few times on
n_clips x 30 s (for rocprofv3 passes).  usage: gpu_run_chain.py [n_clips] [reps] [unfused]"""
import ctypes as C, importlib.util, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("ffi", os.path.join(ROOT, "audio-flow-rs_amd", "_ffi.py"))
ffi = importlib.util.module_from_spec(spec); spec.loader.exec_module(ffi)
lib = ffi.load(os.environ.get("AF_LIB"))
n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
if len(sys.argv) > 3 and sys.argv[3] == "unfused": lib.af_debug_set(b"chain.unfused", 1)
n = 1440000
dev = torch.device("cuda:0")
pcm = (torch.rand(n_clips, n, 2, device=dev) * 2 - 1) * 0.3
k = np.arange(129) - 64
taps = (0.9 * np.sinc(0.9 * k) * np.hamming(129)).astype(np.float32)
cfg = ffi.ChainConfig(); lib.af_chain_default_config(C.byref(cfg))
cfg.fir_taps = taps.ctypes.data; cfg.n_fir_taps = 129
h = C.c_void_p(); assert lib.af_chain_create(C.byref(cfg), C.byref(h)) == 0
tot = C.c_size_t(); assert lib.af_chain_set_batch_uniform(h, n_clips, n, C.byref(tot)) == 0
mel = torch.empty(tot.value, 80, device=dev)
s = torch.cuda.current_stream().cuda_stream
for _ in range(reps): assert lib.af_chain_run_device(h, pcm.data_ptr(), mel.data_ptr(), s) == 0
torch.cuda.synchronize()
if reps >= 30:          # past the clock ramp: time 20 more passes
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): lib.af_chain_run_device(h, pcm.data_ptr(), mel.data_ptr(), s)
    e1.record(); torch.cuda.synchronize()
    print("chain ms", e0.elapsed_time(e1) / 20, "checksum", float(mel.double().sum()))
print("ok", tot.value)
