"""Config 4's chain on one GPU's shard (256 x 30 s stereo 48 kHz), the two launches one after the other against the
pipelined form (chain.hip, run_pipelined) at several chunk counts; same rows bit for bit.
   python3 scripts/gpu_bench_chain_pipeline.py [n_clips]"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import audio_flow_rs_amd as af
if os.environ.get("AF_BENCH_LIB"): af._ffi._lib = af._ffi.load(os.environ["AF_BENCH_LIB"])   # another build (scripts/ablate/build.sh)
dev = torch.device("cuda:0")
n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 256
k = np.arange(129) - 64.0
taps = (0.9 * np.sinc(0.9 * k) * np.hamming(129)).astype(np.float32)
n = 1440000
pcm = (torch.rand(n_clips, n, 2, device=dev) * 2 - 1) * 0.3
s = torch.cuda.current_stream().cuda_stream
def timeit(fn, reps=20):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.25: fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
chain = af.ChainPipeline(fir_taps=taps)
tot = chain.set_batch_uniform(n_clips, n)
mel = torch.empty(tot, 80, device=dev)
out, ref = {}, None
for chunks in [int(c) for c in os.environ.get("AF_CHUNKS", "0,2,4,8,16,32,0,8").split(",")]:
    chain.set_option("pipeline", chunks)
    mel.zero_()
    ms = timeit(lambda: chain.run_device(pcm.data_ptr(), mel.data_ptr(), s))
    if ref is None: ref = mel.clone()
    same = bool(torch.equal(mel, ref))
    alg = pcm.numel() * 4 + mel.numel() * 4
    out.setdefault(str(chunks), []).append({"ms": round(ms, 4), "frac_hbm": round(alg / ms / 1e6 / 8000, 4), "same_bits": same,
                                            "launches": chain.n_launches()})
print(json.dumps(out, indent=1))
