import ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import audio_flow_rs_amd as af
lib = af.library(); dev = torch.device("cuda:0")
k = np.arange(129) - 64.0
taps = (0.9 * np.sinc(0.9 * k) * np.hamming(129)).astype(np.float32)
n = 1440000
pcm = (torch.rand(256, n, 2, device=dev) * 2 - 1) * 0.3
mono16 = (torch.rand(256, 480000, device=dev) * 2 - 1) * 0.3
s = torch.cuda.current_stream().cuda_stream
def timeit(fn, reps=20):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.25: fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for knob in (0, 1, 0, 1):
    lib.af_debug_set(b"features.wg_per_cu", knob)
    chain = af.ChainPipeline(fir_taps=taps)
    tot = chain.set_batch_uniform(256, n)
    mel = torch.empty(tot, 80, device=dev)
    tc = timeit(lambda: chain.run_device(pcm.data_ptr(), mel.data_ptr(), s))
    fp = af.FeaturePipeline(sample_rate=16000, channels=1)
    tf_tot = fp.set_batch_uniform(256, 480000)
    mel2 = torch.empty(tf_tot, 80, device=dev)
    tf = timeit(lambda: fp.run_device(mono16.data_ptr(), mel2.data_ptr(), s))
    print(f"wg_per_cu knob {knob}: chain {tc:.3f} ms, features(mono 16k) alone {tf:.3f} ms, resfir = {tc - tf:.3f} ms")
