"""Ragged batch through af_chain_*: 512 clips of 5 .. 30 s, stereo f32, device-resident.
   python3 scripts/gpu_bench_ragged_chain.py [in_hz ...]   (default 44100 48000)"""
import ctypes as C, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audio_flow_rs_amd as af
dev = torch.device("cuda:0")
k = np.arange(129) - 64.0
taps = (0.9 * np.sinc(0.9 * k) * np.hamming(129)).astype(np.float32)
out = {}
for fi in [int(a) for a in sys.argv[1:]] or [44100, 48000]:
    rng = np.random.default_rng(1)
    lens = rng.integers(5 * fi, 30 * fi, 512)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    pcm = (torch.rand(int(offs[-1]), 2, device=dev) * 2 - 1) * 0.3
    chain = af.ChainPipeline(input_rate=fi, channels=2, output_rate=16000, fir_taps=taps)
    total = chain.set_batch(offs)
    mel = torch.empty(total, 80, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    run = lambda: chain.run_device(pcm.data_ptr(), mel.data_ptr(), s)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3: run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    alg = pcm.numel() * 4 + mel.numel() * 4
    out[str(fi)] = {"clips": 512, "audio_s": float(offs[-1] / fi), "launches": chain.n_launches(), "ms": round(ms, 4),
                    "rtf": round(float(offs[-1] / fi) / (ms * 1e-3)), "frac_hbm": round(alg / ms / 1e6 / 8000, 4)}
print(json.dumps(out, indent=1))
