import numpy as np, torch, sys, time
sys.path.insert(0, ".")
import audio_flow_rs_amd as af
dev = torch.device("cuda:0")
n_clips, n = 256, 1440000
k = np.arange(129) - 64.0
taps = (2 * 7200.0 / 16000.0 * np.sinc(2 * 7200.0 / 16000.0 * k) * np.hamming(129)).astype(np.float32)
pcm = (torch.rand(n_clips, n, 2, device=dev) * 2 - 1) * 0.3
s = torch.cuda.current_stream().cuda_stream
def timeit(fn, reps=20):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3: fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ref = None
for chunk in (256, 64, 32, 16, 8):
    chain = af.ChainPipeline(fir_taps=taps)
    tot = chain.set_batch_uniform(chunk, n)
    nf = tot // chunk
    mel = torch.empty(n_clips * nf, 80, device=dev)
    def run():
        for c in range(0, n_clips, chunk):
            chain.run_device(pcm[c].data_ptr(), mel[c * nf].data_ptr(), s)
    ms = timeit(run)
    if ref is None: ref = mel.clone()
    print(f"chunk {chunk:4d} clips: {ms:.4f} ms per 256 clips, equal to one-shot: {bool(torch.equal(mel, ref))}")
