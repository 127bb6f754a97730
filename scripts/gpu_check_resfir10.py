"""Ten-matrix fused resampler + FIR launch (44.1 k -> 16 k, resfir10_kernel): rows of the fused chain against the stage-by-stage
chain of the same handle type, all four PCM formats, uniform and ragged batches; timing on 256 x 30 s stereo.
   python3 scripts/gpu_check_resfir10.py [--time]"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import audio_flow_rs_amd as af
from util import synth_clip, bits_equal

n129 = np.arange(129) - 64.0
taps = (2 * 7200.0 / 16000.0 * np.sinc(2 * 7200.0 / 16000.0 * n129) * np.hamming(129)).astype(np.float32)
bad = 0
for ch in (() if "--time-only" in sys.argv else (2, 1)):
    for fmt in (0, 1):
        for K in (129, 65, 17):
            t = taps if K == 129 else taps[:K].copy()
            for lens in ([44100 * 3 + 17], [20000, 20000, 20000], [14112 * 2, 14112 * 2], [5, 0, 70001, 30000, 1, 44100, 14113]):
                lens = [n + (n & 1) if (ch == 1 and fmt == 1) else n for n in lens]     # mono i16 clips start on even frames
                clips = [synth_clip(900 + i, n, ch, 44100) for i, n in enumerate(lens)]
                if fmt: clips = [np.clip(np.round(x * 30000.0), -32768, 32767).astype(np.int16) for x in clips]
                x = np.concatenate([c.reshape(-1) for c in clips]) if clips else np.zeros(0, np.float32)
                off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
                rows = []
                for unf in (0, 1):
                    c = af.ChainPipeline(input_rate=44100, channels=ch, output_rate=16000, fir_taps=t, sample_format=fmt)
                    c.set_option("unfused", unf)
                    if len(set(lens)) == 1: c.set_batch_uniform(len(lens), lens[0])
                    else: c.set_batch(off)
                    nl = c.n_launches()
                    rows.append((c.run(x), nl))
                ok = bits_equal(rows[0][0], rows[1][0])
                d = np.abs(rows[0][0].astype(np.float64) - rows[1][0]).max() if rows[0][0].size else 0.0
                print(f"ch {ch} fmt {fmt} K {K} lens {lens[:4]}: launches {rows[0][1]} / {rows[1][1]}  bits_equal {ok}  max |d| {d:.3e}  finite {np.isfinite(rows[0][0]).all()}")
                bad += (not ok) and d > 1e-6
if "--time" in sys.argv or "--time-only" in sys.argv:
    import torch
    dev = torch.device("cuda:0")
    x = (torch.rand(256, 44100 * 30, 2, device=dev) * 2 - 1) * 0.5
    for unf in (0, 1):
        c = af.ChainPipeline(input_rate=44100, channels=2, output_rate=16000, fir_taps=taps)
        c.set_option("unfused", unf)
        tot = c.set_batch_uniform(256, 44100 * 30)
        mel = torch.empty(tot, 80, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        t0 = time.time()
        while time.time() - t0 < 0.3: c.run_device(x.data_ptr(), mel.data_ptr(), s); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): c.run_device(x.data_ptr(), mel.data_ptr(), s)
        e1.record(); torch.cuda.synchronize()
        print(f"256 x 30 s stereo 44.1 kHz chain, unfused={unf}: {c.n_launches()} launches, {e0.elapsed_time(e1) / 20:.4f} ms")
print("FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
