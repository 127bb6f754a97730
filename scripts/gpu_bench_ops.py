"""Timings of the secondary kernels at BASELINE shapes (device-resident data, HIP events on
torch's stream): config 3 resample, FIR (config 4 stage), to_mono, quantise, base64, VAD energy,
plus the PCIe-inclusive rate of the fused feature path through the host-buffer entry point."""
import ctypes as C, os, sys, time, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audio_flow_rs_amd as af
lib = af.library(); dev = torch.device("cuda:0"); P = C.c_void_p
def timeit(fn, k=10, w=2):
    # the chip ramps its clocks for ~25 launches / ~50 ms after idle: warm up past that, whatever w says
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15: fn(); torch.cuda.synchronize()
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
out = {}
# config 3: 4096 x 10 s mono 44.1 kHz -> 16 kHz (7.2 GB in, 2.6 GB out)
nc, nin = 4096, 441000
nout = lib.af_resample_out_len(44100, 16000, nin)
x = torch.rand(nc, nin, device=dev) - 0.5; y = torch.empty(nc, nout, device=dev)
ms = timeit(lambda: lib.af_resample_batch_device(44100, 16000, P(x.data_ptr()), nc, nin, P(y.data_ptr()), None), k=5)
b = (x.numel() + y.numel()) * 4
out["resample_44k1_16k_cfg3"] = dict(ms=ms, alg_GBps=b / ms / 1e6, frac_hbm=b / ms / 1e6 / 8000, rtf=nc * 10 / ms * 1e3)
del x, y
# 48k -> 16k, 1024 x 30 s mono (config 4 stage)
nc, nin = 1024, 1440000
nout = lib.af_resample_out_len(48000, 16000, nin)
x = torch.rand(nc, nin, device=dev) - 0.5; y = torch.empty(nc, nout, device=dev)
ms = timeit(lambda: lib.af_resample_batch_device(48000, 16000, P(x.data_ptr()), nc, nin, P(y.data_ptr()), None), k=5)
b = (x.numel() + y.numel()) * 4
out["resample_48k_16k_cfg4"] = dict(ms=ms, alg_GBps=b / ms / 1e6, frac_hbm=b / ms / 1e6 / 8000)
# FIR K=129 on the 16 kHz clips
taps = torch.rand(129, device=dev) / 129; z = torch.empty_like(y)
ms = timeit(lambda: lib.af_fir_batch_device(P(taps.data_ptr()), 129, P(y.data_ptr()), nc, nout, P(z.data_ptr()), None), k=5)
b = 2 * y.numel() * 4
out["fir129_cfg4"] = dict(ms=ms, alg_GBps=b / ms / 1e6, frac_hbm=b / ms / 1e6 / 8000, tflops=2 * 129 * y.numel() / ms / 1e9)
# a long FIR (K = 1024): six MFMA passes of 176 taps against the thread-per-output kernel
taps_l = torch.rand(1024, device=dev) / 1024
nl = 256
ms = timeit(lambda: lib.af_fir_batch_device(P(taps_l.data_ptr()), 1024, P(y.data_ptr()), nl, nout, P(z.data_ptr()), None), k=3)
out["fir1024_256clips"] = dict(ms=ms, tflops=2 * 1024 * nl * nout / ms / 1e9, note="6 banded MFMA passes")
lib.af_debug_set(b"banded.off", 1)
ms = timeit(lambda: lib.af_fir_batch_device(P(taps_l.data_ptr()), 1024, P(y.data_ptr()), nl, nout, P(z.data_ptr()), None), k=3)
lib.af_debug_set(b"banded.off", 0)
out["fir1024_256clips_valu"] = dict(ms=ms, tflops=2 * 1024 * nl * nout / ms / 1e9, note="thread-per-output kernel")
# quantise / base64 / vad energy on the same buffer
q = torch.empty(y.numel(), dtype=torch.int16, device=dev)
ms = timeit(lambda: lib.af_quantize_pcm16_device(P(y.data_ptr()), y.numel(), P(q.data_ptr()), None))
out["quantize_pcm16"] = dict(ms=ms, alg_GBps=y.numel() * 6 / ms / 1e6, frac_hbm=y.numel() * 6 / ms / 1e6 / 8000)
b64 = torch.empty(lib.af_base64_len(2 * y.numel()) + 16, dtype=torch.uint8, device=dev)
ms = timeit(lambda: lib.af_pcm16_base64_device(P(y.data_ptr()), y.numel(), P(b64.data_ptr()), None))
bb = y.numel() * 4 + b64.numel()
out["pcm16_base64"] = dict(ms=ms, alg_GBps=bb / ms / 1e6, frac_hbm=bb / ms / 1e6 / 8000)
en = torch.empty(y.numel() // 320, device=dev)
ms = timeit(lambda: lib.af_vad_energy_device(P(y.data_ptr()), y.numel() // 320, 320, P(en.data_ptr()), None))
out["vad_energy_320"] = dict(ms=ms, alg_GBps=y.numel() * 4 / ms / 1e6, frac_hbm=y.numel() * 4 / ms / 1e6 / 8000)
# level meter: 50 ms windows (800 samples at 16 kHz) over the same buffer
nw = y.numel() // 800
lv = [torch.empty(nw, device=dev) for _ in range(3)]
ms = timeit(lambda: lib.af_level_meter_device(P(y.data_ptr()), nw, 800, P(lv[0].data_ptr()), P(lv[1].data_ptr()), P(lv[2].data_ptr()), None))
out["level_meter_800"] = dict(ms=ms, alg_GBps=nw * 800 * 4 / ms / 1e6, frac_hbm=nw * 800 * 4 / ms / 1e6 / 8000)
del x, y, z, q, b64, en, lv
# to_mono on config 2's input
s = torch.rand(256 * 1440000, 2, device=dev); m = torch.empty(256 * 1440000, device=dev)
ms = timeit(lambda: lib.af_to_mono_device(P(s.data_ptr()), s.numel(), 2, P(m.data_ptr()), None))
out["to_mono_cfg2"] = dict(ms=ms, alg_GBps=s.numel() * 6 / ms / 1e6, frac_hbm=s.numel() * 6 / ms / 1e6 / 8000)
# fused kernel, power-spectrogram output (AF_OUT_POWER): 256 x 30 s stereo -> [1439232][513]
pp = af.FeaturePipeline(output=af._ffi.AF_OUT_POWER)
tot = pp.set_batch_uniform(256, 1440000)
pw = torch.empty(tot, 513, device=dev)
ms = timeit(lambda: pp.run_device(s.data_ptr(), pw.data_ptr(), torch.cuda.current_stream().cuda_stream), k=5)
bb = s.numel() * 4 + pw.numel() * 4
out["stft_power_cfg2"] = dict(ms=ms, alg_GBps=bb / ms / 1e6, frac_hbm=bb / ms / 1e6 / 8000, frames_per_s=tot / ms * 1e3)
# stand-alone mel contraction on those rows
fb = af.MelFilterbank(sample_rate=48000)
ml = torch.empty(tot, 80, device=dev)
ms = timeit(lambda: fb.apply_device(pw.data_ptr(), tot, ml.data_ptr(), torch.cuda.current_stream().cuda_stream), k=5)
bb = pw.numel() * 4 + ml.numel() * 4
out["mel_apply_cfg2"] = dict(ms=ms, alg_GBps=bb / ms / 1e6, frac_hbm=bb / ms / 1e6 / 8000)
del pw, ml
# config 4 chain (to_mono -> 48k->16k -> FIR 129 -> STFT -> 80 mel), 256 stereo 30 s clips
import numpy as np
taps_h = (np.hamming(129) * np.sinc((np.arange(129) - 64) * 0.9)).astype(np.float32); taps_h /= taps_h.sum()
ch = af.ChainPipeline(input_rate=48000, channels=2, output_rate=16000, fir_taps=taps_h)
totc = ch.set_batch_uniform(256, 1440000)
melc = torch.empty(totc, 80, device=dev)
ms = timeit(lambda: ch.run_device(s.data_ptr(), melc.data_ptr(), torch.cuda.current_stream().cuda_stream), k=5)
bb = s.numel() * 4 + melc.numel() * 4
out["chain_cfg4_256clips"] = dict(ms=ms, alg_GBps=bb / ms / 1e6, frac_hbm=bb / ms / 1e6 / 8000, rtf=256 * 30 / ms * 1e3,
                                  launches=ch.n_launches(),
                                  note="algorithmic bytes = PCM in + mel out")
del melc
# PCIe-inclusive fused features (host buffers), 64 clips
k = 64
pipe = af.FeaturePipeline(); pipe.set_batch_uniform(k, 1440000)
host = s[: k * 1440000].cpu().numpy()
pipe.run(host)
t0 = time.perf_counter(); mel = pipe.run(host); dt = time.perf_counter() - t0
out["features_host_buffers_64clips"] = dict(ms=dt * 1e3, frames_per_s=mel.shape[0] / dt, note="H2D + kernel + D2H, pageable host memory")
# the same with page-locked host buffers (hipHostMalloc / torch pin_memory): hipMemcpy then runs at PCIe speed
try:
    hp = torch.empty(host.size, dtype=torch.float32).pin_memory(); hp.numpy()[:] = host.reshape(-1)
    mp = torch.empty(mel.shape, dtype=torch.float32).pin_memory()
    import ctypes as _C
    def run_pinned():
        st = lib.af_features_run(pipe._h, _C.c_void_p(hp.data_ptr()), hp.numel(), _C.c_void_p(mp.data_ptr()), mp.numel())
        assert st == 0
    run_pinned()
    t0 = time.perf_counter(); run_pinned(); dt = time.perf_counter() - t0
    out["features_host_buffers_64clips_pinned"] = dict(ms=dt * 1e3, frames_per_s=mel.shape[0] / dt, note="H2D + kernel + D2H, page-locked host memory, chunks on three streams")
    # resampler and chain through their host-buffer entry points, page-locked memory
    nr = 64; nri = 1440000; nro = lib.af_resample_out_len(48000, 16000, nri)
    rin = torch.empty(nr * nri, dtype=torch.float32).pin_memory(); rin.numpy()[:] = host.reshape(-1)[: nr * nri]
    rout = torch.empty(nr * nro, dtype=torch.float32).pin_memory()
    def run_res():
        assert lib.af_resample_batch(48000, 16000, _C.c_void_p(rin.data_ptr()), nr, nri, _C.c_void_p(rout.data_ptr()), rout.numel()) == 0
    run_res(); t0 = time.perf_counter(); run_res(); dt2 = time.perf_counter() - t0
    dd = torch.empty(rin.numel(), device=dev); torch.cuda.synchronize()
    dd.copy_(rin); torch.cuda.synchronize(); t0 = time.perf_counter(); dd.copy_(rin); torch.cuda.synchronize(); h2d_s = time.perf_counter() - t0
    do = torch.empty(rout.numel(), device=dev); rout.copy_(do); torch.cuda.synchronize()
    t0 = time.perf_counter(); rout.copy_(do); torch.cuda.synchronize(); d2h_s = time.perf_counter() - t0
    del do
    out["resample_host_buffers_64clips_pinned"] = dict(ms=dt2 * 1e3, h2d_alone_ms=h2d_s * 1e3, d2h_alone_ms=d2h_s * 1e3,
                                                       slower_copy_over_end_to_end=max(h2d_s, d2h_s) / dt2,
                                                       note="64 x 30 s mono 48 -> 16 kHz: H2D + kernel + D2H in clip chunks on three streams")
    del dd
    chp = af.ChainPipeline(input_rate=48000, channels=2, output_rate=16000, fir_taps=taps_h)
    totp = chp.set_batch_uniform(k, 1440000)
    melp = torch.empty(totp * 80, dtype=torch.float32).pin_memory()
    def run_chain():
        assert lib.af_chain_run(chp._h, _C.c_void_p(hp.data_ptr()), hp.numel(), _C.c_void_p(melp.data_ptr()), melp.numel()) == 0
    run_chain(); t0 = time.perf_counter(); run_chain(); dt2 = time.perf_counter() - t0
    dd = torch.empty(hp.numel(), device=dev); dd.copy_(hp); torch.cuda.synchronize(); t0 = time.perf_counter(); dd.copy_(hp); torch.cuda.synchronize(); h2d_s = time.perf_counter() - t0
    out["chain_host_buffers_64clips_pinned"] = dict(ms=dt2 * 1e3, h2d_alone_ms=h2d_s * 1e3, frac_of_h2d=h2d_s / dt2,
                                                    note="64 x 30 s stereo through af_chain_run: one copy in, two launches, one copy out (the mel rows are 5 % of the input)")
    del dd
    # what the copies alone cost on this box: H2D of the same PCM, D2H of the same rows (torch, pinned)
    dbuf = torch.empty(hp.numel(), device=dev); dm = torch.empty(mp.shape, device=dev)
    def h2d(): dbuf.copy_(hp, non_blocking=True); torch.cuda.synchronize()
    def d2h(): mp.copy_(dm, non_blocking=True); torch.cuda.synchronize()
    h2d(); t0 = time.perf_counter(); h2d(); t_h2d = time.perf_counter() - t0
    d2h(); t0 = time.perf_counter(); d2h(); t_d2h = time.perf_counter() - t0
    out["features_host_buffers_64clips_pinned"].update(
        h2d_only_ms=t_h2d * 1e3, d2h_only_ms=t_d2h * 1e3, h2d_GBps=hp.numel() * 4 / t_h2d / 1e9,
        end_to_end_over_h2d_only=t_h2d / dt)
except Exception as e:
    out["features_host_buffers_64clips_pinned"] = dict(ms=0.0, note="failed: %r" % (e,))
print(json.dumps(out, indent=1))
