"""The chain's fused resampler + FIR launch (banded::launch_fused) against the two stand-alone launches
(af_debug_set "chain.unfused" = 1) on the same input: bit for bit where both run the same MFMA step count,
within the docs/OPS.md tolerance otherwise; then the timing of config 4's chain both ways."""
import ctypes as C, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audio_flow_rs_amd as af
lib = af.library(); dev = torch.device("cuda:0")


def taps_of(K):
    n = np.arange(K) - (K - 1) / 2
    return (0.9 * np.sinc(0.9 * n) * np.hamming(K)).astype(np.float32)


def run(chain, pcm, total, unfused):
    lib.af_debug_set(b"chain.unfused", 1 if unfused else 0)
    mel = torch.full((max(total, 1), 80), float("nan"), device=dev)
    chain.run_device(pcm.data_ptr(), mel.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    lib.af_debug_set(b"chain.unfused", 0)
    return mel[:total]


bad = 0
for ch, fi, fo, K, n_clips, n in [(2, 48000, 16000, 129, 3, 40000), (1, 48000, 16000, 129, 3, 40000),
                                  (2, 48000, 16000, 129, 2, 1440000), (2, 48000, 16000, 129, 5, 123457),
                                  (1, 32000, 16000, 129, 3, 70001), (2, 48000, 24000, 129, 3, 50000),
                                  (2, 48000, 16000, 65, 3, 40000), (2, 48000, 16000, 33, 3, 40000),
                                  (2, 48000, 16000, 100, 3, 40000), (1, 48000, 16000, 1, 2, 40000),
                                  (2, 48000, 16000, 129, 1, 11520 * 3 + 5), (2, 48000, 16000, 129, 300, 50000)]:
    chain = af.ChainPipeline(input_rate=fi, channels=ch, output_rate=fo, fir_taps=taps_of(K))
    total = chain.set_batch_uniform(n_clips, n)
    g = torch.Generator(device=dev); g.manual_seed(n + K)
    pcm = torch.rand(n_clips, n, ch, device=dev, generator=g) - 0.5
    a = run(chain, pcm, total, False)
    b = run(chain, pcm, total, True)
    eq = bool(torch.equal(a, b))
    err = float(((a - b).norm(dim=1) / b.norm(dim=1).clamp_min(1e-30)).max()) if total else 0.0
    fin = bool(torch.isfinite(a).all())
    ok = fin and (eq if K == 129 else err < 2e-6)
    bad += 0 if ok else 1
    print(f"ch={ch} {fi}->{fo} K={K} clips={n_clips} n={n} rows={total}: equal={eq} max_rel={err:.2e} finite={fin} {'ok' if ok else 'FAIL'}",
          flush=True)

# timing: BASELINE config 4 per-GPU share in bench.py's size (256 x 30 s stereo 48 kHz)
n_clips, n = 256, 1440000
chain = af.ChainPipeline(input_rate=48000, channels=2, output_rate=16000, fir_taps=taps_of(129))
total = chain.set_batch_uniform(n_clips, n)
pcm = torch.rand(n_clips, n, 2, device=dev) - 0.5
mel = torch.empty(total, 80, device=dev)
st = torch.cuda.current_stream().cuda_stream
out = {}
for name, knob in (("fused", 0), ("unfused", 1), ("fused_again", 0)):
    lib.af_debug_set(b"chain.unfused", knob)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        chain.run_device(pcm.data_ptr(), mel.data_ptr(), st); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): chain.run_device(pcm.data_ptr(), mel.data_ptr(), st)
    e1.record(); torch.cuda.synchronize()
    out[name] = e0.elapsed_time(e1) / 20
lib.af_debug_set(b"chain.unfused", 0)
alg = pcm.numel() * 4 + mel.numel() * 4
for k, v in out.items(): print(f"chain {k}: {v:.3f} ms  {alg / v / 1e6:.0f} GB/s  frac {alg / v / 1e6 / 8000:.3f}")
print(json.dumps(out))
sys.exit(1 if bad else 0)
