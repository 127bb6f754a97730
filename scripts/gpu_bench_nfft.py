"""Frame lengths outside {1024, 512, 256} (features_dft.hip, the DFT as a matrix product): time and matrix-pipe rate.
   python3 scripts/gpu_bench_nfft.py [--clips 256] [--seconds 30] [--sr 16000] [--n_fft 400] [--hop 160] [--json PATH]
Algorithmic work per frame: 2 (re, im) x n_fft x (n_fft / 2 + 1) multiply-adds for the transform + the mel product on its band
tiles; the rate is quoted on the transform's multiply-adds alone against the dense f32 matrix peak (157.3 TFLOP/s)."""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audio_flow_rs_amd as af
ap = argparse.ArgumentParser()
ap.add_argument("--clips", type=int, default=256); ap.add_argument("--seconds", type=float, default=30.0)
ap.add_argument("--sr", type=int, default=16000); ap.add_argument("--n_fft", type=int, default=400)
ap.add_argument("--hop", type=int, default=160); ap.add_argument("--channels", type=int, default=1)
ap.add_argument("--reps", type=int, default=10); ap.add_argument("--json", default="")
a = ap.parse_args()
dev = torch.device("cuda:0")
n = int(a.seconds * a.sr)
x = (torch.rand(a.clips, n, a.channels, device=dev) * 2 - 1) * 0.3
p = af.FeaturePipeline(sample_rate=a.sr, channels=a.channels, n_fft=a.n_fft, hop=a.hop)
tot = p.set_batch_uniform(a.clips, n)
mel = torch.empty(tot, 80, device=dev)
s = torch.cuda.current_stream().cuda_stream
t0 = time.time()
while time.time() - t0 < 0.5: p.run_device(x.data_ptr(), mel.data_ptr(), s); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.reps): p.run_device(x.data_ptr(), mel.data_ptr(), s)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.reps
n_bins = a.n_fft // 2 + 1
kpad, npad = (a.n_fft + 15) // 16 * 16, (n_bins + 15) // 16 * 16
flop = 2.0 * 2 * a.n_fft * n_bins * tot
flop_issued = 2.0 * 2 * kpad * npad * ((tot + 15) // 16 * 16)
r = {"workload": f"{a.clips} x {a.seconds:g} s, {a.channels} ch, {a.sr} Hz, n_fft {a.n_fft}, hop {a.hop}, 80 mel", "frames": tot, "ms": round(ms, 4),
     "frames_per_s": round(tot / ms * 1e3), "rtf": round(a.clips * a.seconds / (ms * 1e-3)),
     "tflops_algorithmic": round(flop / ms * 1e-9, 2), "tflops_issued": round(flop_issued / ms * 1e-9, 2),
     "frac_of_f32_matrix_peak_157p3": round(flop / ms * 1e-9 / 157.3, 3),
     "hbm_bytes": x.numel() * 4 + tot * 80 * 4, "hbm_TBps": round((x.numel() * 4 + tot * 80 * 4) / ms * 1e-9, 3)}
print(json.dumps(r))
if a.json: open(a.json, "w").write(json.dumps(r, indent=1) + "\n")
