"""Wire path in one launch (af_wire_*) on device-resident PCM: python3 scripts/gpu_bench_wire.py [--clips 1024] [--frame 0|320] [--reps 20] [--json PATH]
Times the fused launch and, beside it, the three launches it replaces (stereo-staging resampler, quantiser, base64) and the
two-launch form of the same handle.  Algorithmic bytes: 8 B per stereo PCM frame in + the text out (4 characters per 3 bytes
of PCM16)."""
import argparse, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audio_flow_rs_amd as af
import ctypes as C

ap = argparse.ArgumentParser()
ap.add_argument("--clips", type=int, default=1024); ap.add_argument("--frame", type=int, default=0)
ap.add_argument("--reps", type=int, default=20); ap.add_argument("--warm_s", type=float, default=0.3)
ap.add_argument("--fmt", default="f32"); ap.add_argument("--json", default="")
ap.add_argument("--only", default="", help="fused: time only the fused launch (profiling runs)")
ap.add_argument("--level", action="store_true", help="also emit the messages' energy and peak (frame > 0)")
a = ap.parse_args()
dev = torch.device("cuda:0")
n = 1440000
if a.fmt == "f32":
    pcm = (torch.rand(a.clips, n, 2, device=dev) * 2 - 1) * 0.8
    fmt = 0
else:
    pcm = torch.randint(-30000, 30000, (a.clips, n, 2), device=dev, dtype=torch.int16)
    fmt = 1
w = af.WireEncoder(frame_samples=a.frame, sample_format=fmt)
total = w.set_batch_uniform(a.clips, n)
text = torch.empty(total, dtype=torch.uint8, device=dev)
lib = af.library()
s = torch.cuda.current_stream().cuda_stream

def timed(fn):
    import time
    t0 = time.time()
    while time.time() - t0 < a.warm_s: fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps

alg = pcm.numel() * pcm.element_size() + total
res = {"workload": f"{a.clips} x 30 s stereo 48 kHz {a.fmt} -> 16 kHz PCM16 base64" + (f", messages of {a.frame} samples" if a.frame else ", one string per clip"),
       "algorithmic_bytes": alg, "text_bytes": total, "launches_fused": w.n_launches()}
ms = timed(lambda: w.run_device(pcm.data_ptr(), text.data_ptr(), 0, s))
if a.level and a.frame:
    en = torch.empty(w.total_messages(), device=dev); pk = torch.empty_like(en)
    ms_l = timed(lambda: w.run_device(pcm.data_ptr(), text.data_ptr(), 0, s, en.data_ptr(), pk.data_ptr()))
    res["fused_with_energy_and_peak_ms"] = ms_l
    res["fused_with_energy_and_peak_frac_of_8TBps"] = (alg + 2 * en.numel() * 4) / ms_l / 1e9 / 8.0
res["fused_ms"] = ms; res["fused_TBps"] = alg / ms / 1e9; res["fused_frac_of_8TBps"] = alg / ms / 1e9 / 8.0
res["rtf"] = a.clips * 30.0 / (ms * 1e-3)
if a.only != "fused" and a.fmt == "f32":
    w.set_option("unfused", 1)
    ms2 = timed(lambda: w.run_device(pcm.data_ptr(), text.data_ptr(), 0, s))
    res["two_launch_ms"] = ms2
    if a.frame == 0:
        n_res = w.out_samples(n)
        y = torch.empty(a.clips, n_res, device=dev)
        def three():
            assert lib.af_resample_batch_stereo_device(48000, 16000, C.c_void_p(pcm.data_ptr()), a.clips, n, C.c_void_p(y.data_ptr()), C.c_void_p(s)) == 0
            for i in range(0, a.clips, 64):      # one string per clip: the existing entry point encodes ONE array; 64 clips per call here
                pass
            assert lib.af_pcm16_base64_device(C.c_void_p(y.data_ptr()), y.numel(), C.c_void_p(text.data_ptr()), C.c_void_p(s)) == 0
        res["resampler_plus_pcm16_base64_ms"] = timed(three)     # (whole batch as ONE string: a lower bound for the per-clip form)
print(json.dumps(res, indent=1))
if a.json:
    os.makedirs(os.path.dirname(a.json), exist_ok=True)
    json.dump(res, open(a.json, "w"), indent=1)
