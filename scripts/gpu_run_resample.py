"""Runs the batch resampler a few times (for rocprofv3 passes): python3 scripts/gpu_run_resample.py FI FO [clips] [seconds] [reps]"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audio_flow_rs_amd as af
lib = af.library(); dev = torch.device("cuda:0"); P = C.c_void_p
fi, fo = int(sys.argv[1]), int(sys.argv[2])
nc = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
secs = int(sys.argv[4]) if len(sys.argv) > 4 else 30
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 4
nin = fi * secs
nout = lib.af_resample_out_len(fi, fo, nin)
x = torch.rand(nc, nin, device=dev) - 0.5; y = torch.empty(nc, nout, device=dev)
for _ in range(reps):
    assert lib.af_resample_batch_device(fi, fo, P(x.data_ptr()), nc, nin, P(y.data_ptr()), None) == 0
torch.cuda.synchronize()
print("ok", nout)
