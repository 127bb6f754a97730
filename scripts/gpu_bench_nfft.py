"""Frame lengths outside {1024, 512, 256} (features_dft.hip, the DFT as a matrix product): time and matrix-pipe rate.
   python3 scripts/gpu_bench_nfft.py [--clips 256] [--seconds 30] [--sr 16000] [--n_fft 400] [--hop 160] [--json PATH]
The kernel issues 2 (re, im) x (n_fft / 2) x (n_fft / 2 + 1) multiply-adds per frame (the symmetric window halves the DFT sum's
length; padded to tiles of 16) + the mel product on its band tiles; `frac_of_f32_matrix_peak` is that issue rate against the
dense f32 matrix peak (157.3 TFLOP/s), `tflops_of_the_dft_as_defined` the rate of the unfolded sum it replaces."""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audio_flow_rs_amd as af
ap = argparse.ArgumentParser()
ap.add_argument("--clips", type=int, default=256); ap.add_argument("--seconds", type=float, default=30.0)
ap.add_argument("--sr", type=int, default=16000); ap.add_argument("--n_fft", type=int, default=400)
ap.add_argument("--hop", type=int, default=160); ap.add_argument("--channels", type=int, default=1)
ap.add_argument("--reps", type=int, default=10); ap.add_argument("--waves", type=int, default=0); ap.add_argument("--wg_per_cu", type=int, default=0); ap.add_argument("--queue", type=int, default=-1)
ap.add_argument("--json", default=""); ap.add_argument("--stamps", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
n = int(a.seconds * a.sr)
x = (torch.rand(a.clips, n, a.channels, device=dev) * 2 - 1) * 0.3
p = af.FeaturePipeline(sample_rate=a.sr, channels=a.channels, n_fft=a.n_fft, hop=a.hop)
if a.wg_per_cu: assert af.library().af_features_set_option(p._h, b"wg_per_cu", a.wg_per_cu) == 0
if a.queue >= 0: assert af.library().af_features_set_option(p._h, b"dft_queue", a.queue) == 0
if a.waves: assert af.library().af_features_set_option(p._h, b"dft_waves", a.waves) == 0
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
kg, nb = max((a.n_fft // 2 + 15) // 16, 4), (n_bins + 15) // 16
flop_def = 2.0 * 2 * a.n_fft * n_bins * tot                       # the DFT sum as defined: n_fft x n_bins complex terms per frame
flop_mfma = 2.0 * 2 * (16 * kg) * (16 * nb) * ((tot + 15) // 16 * 16)   # what the kernel issues: the window's symmetry halves the length
r = {"workload": f"{a.clips} x {a.seconds:g} s, {a.channels} ch, {a.sr} Hz, n_fft {a.n_fft}, hop {a.hop}, 80 mel", "frames": tot, "ms": round(ms, 4),
     "frames_per_s": round(tot / ms * 1e3), "rtf": round(a.clips * a.seconds / (ms * 1e-3)),
     "tflops_issued_f32_mfma": round(flop_mfma / ms * 1e-9, 2), "frac_of_f32_matrix_peak_157p3": round(flop_mfma / ms * 1e-9 / 157.3, 3),
     "tflops_of_the_dft_as_defined": round(flop_def / ms * 1e-9, 2),
     "hbm_bytes": x.numel() * 4 + tot * 80 * 4, "hbm_TBps": round((x.numel() * 4 + tot * 80 * 4) / ms * 1e-9, 3)}
if "--stamps" in sys.argv:
    import numpy as np, ctypes as C
    lib = af.library()
    assert lib.af_features_debug_stamps(p._h, 1, None) == 0
    for _ in range(5): p.run_device(x.data_ptr(), mel.data_ptr(), s)
    torch.cuda.synchronize()
    st = np.zeros(32, np.uint64)
    assert lib.af_features_debug_stamps(p._h, 0, st.ctypes.data) == 0
    u = float(st[2])
    r["stamps"] = {"stage_cycles_per_unit": round(float(st[0]) / u), "product_cycles_per_unit": round(float(st[1]) / u),
                   "clock_GHz": round(float(st[3]) / float(st[4]) * 0.1, 3), "waves_per_launch": int(st[5]) // 5,
                   "lifetime_cycles_per_wave": round(float(st[3]) / float(st[5]))}
print(json.dumps(r))
if a.json: open(a.json, "w").write(json.dumps(r, indent=1) + "\n")
