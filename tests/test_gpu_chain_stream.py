"""GPU parity of the chained path (BASELINE config 4: to_mono -> 48->16 kHz polyphase ->
FIR -> STFT -> mel) and of chunked streaming with a carried halo and a hipGraph-captured
step (BASELINE config 5).  Streaming == one-shot is bit-exact; the chain is compared with
the float64 oracle chain at the docs/OPS.md tolerance."""
import numpy as np
import pytest

from util import bits_equal, frame_rel_err, synth_clip

pytestmark = pytest.mark.gpu
TOL = 1e-5


def lowpass_taps(K=129, fc=7200.0, sr=16000.0):
    n = np.arange(K) - (K - 1) / 2
    return (2 * fc / sr * np.sinc(2 * fc / sr * n) * np.hamming(K)).astype(np.float32)


def oracle_chain(po, pcm, ch, fi, fo, taps, hop=256, n_mels=80):
    mono = po.to_mono(pcm, ch)
    y = po.resample(mono, fi, fo).astype(np.float32) if fi != fo else mono
    if taps is not None:
        y = po.fir(y, taps).astype(np.float32)
    return po.features(y, 1, float(fo), hop=hop, n_mels=n_mels)


@pytest.mark.parametrize("ch,fi,fo,with_fir", [(2, 48000, 16000, True), (1, 44100, 16000, False),
                                               (2, 48000, 48000, True), (1, 16000, 16000, False)])
def test_chain_vs_oracle(af, po, ch, fi, fo, with_fir):
    taps = lowpass_taps() if with_fir else None
    n_clips, n = 3, 40000
    clips = [synth_clip(90 + i, n, ch, fi) for i in range(n_clips)]
    chain = af.ChainPipeline(input_rate=fi, channels=ch, output_rate=fo, fir_taps=taps)
    total = chain.set_batch_uniform(n_clips, n)
    nf = chain.n_frames(n)
    assert total == n_clips * nf
    assert nf == po.stft_n_frames(po.resample_out_len(n, *po.resample_ratio(fi, fo)))   # exact
    mel = chain.run(np.concatenate(clips))
    for i, x in enumerate(clips):
        ref = oracle_chain(po, x, ch, fi, fo, taps)
        # the f32 intermediates (resampler, FIR) add their own 1e-6-level rounding
        assert frame_rel_err(mel[i * nf:(i + 1) * nf], ref) < 2e-5


@pytest.mark.parametrize("ch,fi,fo,K,n_clips,n", [
    (2, 48000, 16000, 129, 3, 40000),      # config 4's stages
    (1, 48000, 16000, 129, 2, 123457),     # mono, ragged length
    (2, 48000, 16000, 129, 1, 11520 * 3 + 5),   # a few samples into the fourth tile
    (2, 48000, 16000, 129, 300, 50000),    # more tiles than workgroups
    (1, 32000, 16000, 129, 3, 70001), (2, 48000, 24000, 129, 3, 50000),
    (2, 48000, 16000, 100, 3, 40000),      # history not a multiple of 16: padded segments
    (2, 48000, 16000, 66, 2, 40000), (2, 48000, 16000, 65, 2, 40000), (1, 48000, 16000, 33, 2, 40000),
    (1, 48000, 16000, 1, 2, 40000),
    # tile edges of the fused launch (3840 FIR outputs = 11520 input frames per tile): exact, one over, one under
    (2, 48000, 16000, 129, 2, 2 * 11520), (2, 48000, 16000, 129, 2, 2 * 11520 + 3), (2, 48000, 16000, 129, 2, 2 * 11520 - 3),
    (1, 48000, 16000, 129, 3, 3100),       # shorter than the FIR history + one frame
    (2, 32000, 16000, 129, 2, 2 * 7680 + 2)])
def test_chain_fused_stage_equals_separate_launches(af, ch, fi, fo, K, n_clips, n):
    """The one-launch downmix + resampler + FIR (banded::launch_fused) against the stand-alone kernels
    (af_debug_set "chain.unfused"): the same block alignment and MFMA step order, so bit for bit whenever the
    FIR chains have the same step layout (34..65 and 98..129 taps: 16/20 and 32/36 steps differ only in
    all-zero tail steps); other lengths are judged on the per-frame error."""
    torch = pytest.importorskip("torch")
    import ctypes as C
    lib = af.library()
    info = (C.c_int32 * 10)()
    assert lib.af_chain_fused_plan_describe(fi, fo, ch, K, info) == 1
    dev = torch.device("cuda:0")
    k = np.arange(K) - (K - 1) / 2
    taps = (0.9 * np.sinc(0.9 * k) * np.hamming(K)).astype(np.float32)
    chain = af.ChainPipeline(input_rate=fi, channels=ch, output_rate=fo, fir_taps=taps)
    total = chain.set_batch_uniform(n_clips, n)
    gen = torch.Generator(device=dev)
    gen.manual_seed(K + n)
    pcm = torch.rand(n_clips, n, ch, device=dev, generator=gen) - 0.5
    stream = torch.cuda.current_stream().cuda_stream
    out = []
    try:
        for unfused in (0, 1):
            assert lib.af_debug_set(b"chain.unfused", unfused) == 0
            mel = torch.full((total, 80), float("nan"), device=dev)
            chain.run_device(pcm.data_ptr(), mel.data_ptr(), stream)
            torch.cuda.synchronize()
            out.append(mel)
    finally:
        lib.af_debug_set(b"chain.unfused", 0)
    assert bool(torch.isfinite(out[0]).all())
    assert chain.n_launches() == 2
    if 34 <= K <= 65 or 98 <= K <= 129 or K == 1:
        assert bits_equal(out[0], out[1])
    else:
        rel = (out[0] - out[1]).norm(dim=1) / out[1].norm(dim=1).clamp_min(1e-30)
        assert float(rel.max()) < 2e-6


def test_chain_config4_per_gpu_share(af, po):
    # BASELINE config 4's share of one GPU: 1024 clips x 30 s stereo 48 kHz (11.8 GB) through the whole
    # chain; every row written, duplicates identical, determinism, sampled clips against the oracle chain
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    n_clips, n = 1024, 1440000
    taps = lowpass_taps()
    chain = af.ChainPipeline(input_rate=48000, channels=2, output_rate=16000, fir_taps=taps)
    total = chain.set_batch_uniform(n_clips, n)
    nf = chain.n_frames(n)
    assert nf == po.stft_n_frames(480000) == 1872 and total == n_clips * nf
    gen = torch.Generator(device=dev)
    gen.manual_seed(4)
    pcm = torch.empty(n_clips, n, 2, device=dev)
    for c0 in range(0, n_clips, 128):                               # in slices: torch.rand has no in-place form
        pcm[c0:c0 + 128] = torch.rand(128, n, 2, device=dev, generator=gen).mul_(0.6).sub_(0.3)
    pcm[1023] = pcm[5]
    mel = torch.full((total, 80), float("nan"), device=dev)
    mel2 = torch.empty_like(mel)
    stream = torch.cuda.current_stream().cuda_stream
    chain.run_device(pcm.data_ptr(), mel.data_ptr(), stream)
    chain.run_device(pcm.data_ptr(), mel2.data_ptr(), stream)
    torch.cuda.synchronize()
    m = mel.view(n_clips, nf, 80)
    assert bool(torch.isfinite(mel).all()) and bits_equal(mel, mel2) and bits_equal(m[1023], m[5])
    for c in (0, 512, 1022):
        ref = oracle_chain(po, pcm[c].cpu().numpy().reshape(-1), 2, 48000, 16000, taps)
        assert frame_rel_err(m[c].cpu().numpy(), ref) < 2e-5


def test_chain_rejects_bad_config(af):
    with pytest.raises(af.AudioError):
        af.ChainPipeline(channels=3)
    with pytest.raises(af.AudioError):
        af.ChainPipeline(input_rate=0)
    with pytest.raises(af.AudioError):
        af.ChainPipeline(fir_taps=np.zeros(2000, np.float32))


def test_streaming_chunks_equal_one_shot_bit_exact(af, po):
    hop, chunk = 256, 16 * 256 * 3
    x = synth_clip(95, 100000, 2, 48000)
    one = af.FeaturePipeline()
    n_one = one.set_batch_uniform(1, 100000)
    full = one.run(x)
    st = af.FeaturePipeline()
    starts, lengths = af.FeaturePipeline.stream_spans(100000, chunk)
    n_st = st.set_batch_spans(starts, lengths)
    assert n_st == n_one == po.stft_n_frames(100000)            # same frames, same order
    chunked = st.run(x)
    assert bits_equal(chunked, full)                         # bit-exact
    assert frame_rel_err(full, po.features(x, 2, 48000.0)) < TOL


def test_streaming_step_by_step_with_hipgraph(af, po):
    """Config 5 shape in miniature: a long stereo stream processed in fixed-size steps; the
    steady-state step (one fused-kernel launch on a chunk + halo) is captured into a hipGraph
    once and replayed, the input window sliding through a staging buffer."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    hop, n_fft = 256, 1024
    chunk = 16 * hop * 8                                        # 32768 frames per step
    n_steps = 6
    n_total = chunk * n_steps + (n_fft - hop)
    x = synth_clip(96, n_total, 2, 48000)
    stream_dev = torch.from_numpy(x).to(dev).view(n_total, 2)
    span = chunk + n_fft - hop
    pipe = af.FeaturePipeline()
    frames_per_step = pipe.set_batch_uniform(1, span)
    assert frames_per_step == chunk // hop
    stage = torch.zeros(span, 2, device=dev)
    out = torch.empty(frames_per_step, 80, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        stage.copy_(stream_dev[0:span])
        pipe.run_device(stage.data_ptr(), out.data_ptr(), s.cuda_stream)     # warm-up (sets attrs)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            pipe.run_device(stage.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    mels = []
    for k in range(n_steps):
        stage.copy_(stream_dev[k * chunk:k * chunk + span])
        g.replay()
        torch.cuda.synchronize()
        mels.append(out.cpu().numpy().copy())
    got = np.concatenate(mels)
    one = af.FeaturePipeline()
    one.set_batch_uniform(1, n_total)
    full = one.run(x)
    assert got.shape == full.shape == (n_steps * frames_per_step, 80)
    assert bits_equal(got, full)                             # same frame indices, same bits


def test_chain_is_capturable_without_a_warm_up(af, po):
    """af_chain_run_device only launches kernels (tables, FIR fragments and intermediates exist
    after create / set_batch), so its very first call can be captured into a hipGraph."""
    torch = pytest.importorskip("torch")
    taps = lowpass_taps()
    n_clips, n = 2, 48000
    clips = [synth_clip(400 + i, n, 2, 48000) for i in range(n_clips)]
    chain = af.ChainPipeline(input_rate=48000, channels=2, output_rate=16000, fir_taps=taps)
    total = chain.set_batch_uniform(n_clips, n)
    dev = torch.device("cuda:0")
    pcm = torch.from_numpy(np.concatenate(clips)).to(dev)
    mel = torch.zeros(total, 80, device=dev)
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            chain.run_device(pcm.data_ptr(), mel.data_ptr(), s.cuda_stream)
    g.replay()
    torch.cuda.synchronize()
    nf = chain.n_frames(n)
    got = mel.cpu().numpy()
    for i, x in enumerate(clips):
        assert frame_rel_err(got[i * nf:(i + 1) * nf], oracle_chain(po, x, 2, 48000, 16000, taps)) < 2e-5


# ---- FeatureStream (af_stream_*): library-owned staging, halo carry and hipGraph step -----------
def test_feature_stream_small_steps_vs_one_shot_and_oracle(af, po):
    # ragged pushes over several steps, a partial last step, i16 and mono variants
    for ch, fmt, n in [(2, 0, 16 * 256 * 7 + 1234), (1, 1, 16 * 256 * 3 + 77), (2, 0, 5000), (1, 0, 700)]:
        x = synth_clip(300 + ch + fmt, n, ch, 48000)
        if fmt:
            x = np.round(x * 32767.0).astype(np.int16)
        st = af.FeatureStream(step_frames=16 * 256 * 2, channels=ch, sample_format=fmt)
        rng = np.random.default_rng(5)
        rows, pos = [], 0
        while pos < n:
            k = int(rng.integers(0, 9000))
            rows.append(st.push(x[pos * ch:(pos + k) * ch]))
            pos += k
        rows.append(st.flush())
        got = np.concatenate(rows)
        assert got.shape[0] == st.total_rows(n) == po.stft_n_frames(n)          # bookkeeping exact
        one = af.FeaturePipeline(channels=ch, sample_format=fmt)
        one.set_batch_uniform(1, n)
        assert bits_equal(got, one.run(x))                                    # bit for bit
        if got.shape[0]:
            xf = x.astype(np.float32) / np.float32(32768.0) if fmt else x
            assert frame_rel_err(got, po.features(xf, ch, 48000.0)) < TOL
        with pytest.raises(af.AudioError):
            st.push(x[:8 * ch])                                                   # pushed after flush
        st.reset()
        assert st.push(x[:100 * ch]).shape[0] == 0


def test_feature_stream_ten_minutes_bit_equal_to_one_shot(af):
    """BASELINE config 5 through the product API: >= 10 min of 48 kHz stereo in fixed 2^20-frame
    steps (one hipGraph replay each) == one-shot framing of the same signal, bit for bit."""
    torch = pytest.importorskip("torch")
    sr, ch, step = 48000, 2, 1 << 20
    n = 10 * 60 * sr + 12345                                     # 28.8 M frames + a partial step
    gen = torch.Generator().manual_seed(11)
    x = (torch.rand(n, ch, generator=gen) * 2 - 1).mul_(0.4).numpy()           # 230 MB f32
    st = af.FeatureStream(step_frames=step)
    rows, dev_ms, wall_ms = [], [], []
    for pos in range(0, n, 300000):                              # feeder pushes ~6 s at a time
        r = st.push(x[pos:pos + 300000])
        if r.shape[0]:
            s = st.stats()
            dev_ms.append(s["device_ms"]); wall_ms.append(s["wall_ms"])
        rows.append(r)
    rows.append(st.flush())
    got = np.concatenate(rows)
    assert got.shape == (st.total_rows(n), 80) and st.stats()["steps"] == n // step == 27
    one = af.FeaturePipeline()
    one.set_batch_uniform(1, n)
    full = one.run(x.reshape(-1))
    assert bits_equal(got, full)
    # per-step latency (device time of the replayed graph incl. both PCIe copies, and host wall time)
    print(f"\nfeature stream: {len(dev_ms)} steps of {step / sr:.1f} s audio; device ms/step "
          f"median {np.median(dev_ms):.2f} max {np.max(dev_ms):.2f}; wall ms/step median {np.median(wall_ms):.2f}; "
          f"real-time factor {step / sr / (np.median(wall_ms) * 1e-3):.0f}x")
    assert np.median(wall_ms) * 1e-3 < step / sr / 100            # far faster than real time


def test_feature_stream_eight_hours_config5(af):
    """BASELINE config 5 at its full length: ONE 8 h stereo 48 kHz stream (1.3824e9 frames, 11 GB of f32)
    through af_stream_* in 2^20-frame steps (1318 hipGraph replays + the flush) against one-shot framing of
    the same signal on the device, bit for bit; the 8-GPU form of the config shards this stream in time
    (shard.stream_shards), every rank running the same loop on its span."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    sr, step = 48000, 1 << 20
    n = 8 * 3600 * sr
    piece = 4 * step                                                  # the feeder hands over 87 s at a time
    gen = torch.Generator().manual_seed(2024)
    base = torch.rand(piece + 7919, 2, generator=gen).mul_(0.6).sub_(0.3)     # 34 MB of noise, reused at
    t = torch.arange(piece, dtype=torch.float32)                               # shifting offsets + a slow sweep
    st = af.FeatureStream(step_frames=step)
    one = af.FeaturePipeline()
    total = one.set_batch_uniform(1, n)
    assert total == st.total_rows(n) == 1 + (n - 1024) // 256 == 5399997
    pcm = torch.empty(n, 2, device=dev)                               # the same stream, resident for the one-shot run
    rows = torch.empty(total, 80)
    pos = row = 0
    wall = []
    k = 0
    while pos < n:
        m = min(piece, n - pos)
        off = (k * 977) % 7919
        x = base[off:off + m].clone()
        x[:, 0] += 0.2 * torch.sin(t[:m] * (2 * np.pi * (200.0 + 3.0 * k) / sr))
        pcm[pos:pos + m].copy_(x)
        r = st.push(x.numpy())
        rows[row:row + r.shape[0]] = torch.from_numpy(r)
        row += r.shape[0]
        if r.shape[0]:
            wall.append(st.stats()["wall_ms"])
        pos += m
        k += 1
    tail = st.flush()
    rows[row:row + tail.shape[0]] = torch.from_numpy(tail)
    row += tail.shape[0]
    assert row == total and st.stats()["steps"] == (n - 768) // step == 1318
    full = torch.empty(total, 80, device=dev)
    one.run_device(pcm.data_ptr(), full.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert bits_equal(rows, full.cpu())
    print(f"\n8 h stream: {st.stats()['steps']} steps, wall ms per step median {np.median(wall):.2f}, "
          f"{n / sr / (np.median(wall) * 1e-3 * st.stats()['steps']):.0f}x real time on the step path alone")


# ---- ragged batches and i16 PCM through af_chain_* (capture.rs:13-28, websocket.rs:246-251) ----------------------
def _as_i16(x):
    return np.round(np.asarray(x) * 32767.0).astype(np.int16)


@pytest.mark.parametrize("ch,fmt", [(2, 0), (1, 0), (2, 1), (1, 1)])
def test_chain_ragged_and_i16_fused_equals_unfused_and_oracle(af, po, ch, fmt):
    """A ragged batch (lengths from empty to several tiles, one shorter than n_fft at 16 kHz) in both PCM formats:
    the fused launch (one record per tile) == the clip-by-clip stage path bit for bit (129 taps: same MFMA step
    layout), both within tolerance of the chained oracle, frame counts exact."""
    taps = lowpass_taps()
    lens = [40000, 0, 11520 * 2, 3000, 123456, 11520 * 2 + 2, 70, 50001 + (ch == 1 and fmt == 1)]
    if ch == 1 and fmt == 1:
        lens = [n + (n & 1) for n in lens]              # mono i16: even clip offsets keep the fused launch
    clips = [synth_clip(700 + i, n, ch, 48000) for i, n in enumerate(lens)]
    pcm = np.concatenate([c for c in clips if c.size] or [np.zeros(0, np.float32)])
    if fmt:
        pcm = _as_i16(pcm)
    off = np.concatenate([[0], np.cumsum(lens)])
    chain = af.ChainPipeline(input_rate=48000, channels=ch, output_rate=16000, fir_taps=taps, sample_format=fmt)
    total = chain.set_batch(off)
    nfs = [chain.n_frames(n) for n in lens]
    assert total == sum(nfs) and nfs == [po.stft_n_frames(po.resample_out_len(n, 1, 3)) for n in lens]
    assert chain.n_launches() == 2
    fused = chain.run(pcm)
    chain.set_option("unfused", 1)
    assert chain.n_launches() > 2
    staged = chain.run(pcm)
    chain.set_option("unfused", -1)
    assert bits_equal(fused, staged)
    row = 0
    for x, nf in zip(clips, nfs):
        if nf:
            xf = _as_i16(x).astype(np.float32) / np.float32(32768.0) if fmt else x
            assert frame_rel_err(fused[row:row + nf], oracle_chain(po, xf, ch, 48000, 16000, taps)) < 2e-5
        row += nf
    # the same clips as a uniform batch of one length: rows of equal clips agree bit for bit across batch shapes
    uni = af.ChainPipeline(input_rate=48000, channels=ch, output_rate=16000, fir_taps=taps, sample_format=fmt)
    uni.set_batch_uniform(1, lens[4])
    x4 = clips[4] if not fmt else _as_i16(clips[4])
    r0 = sum(nfs[:4])
    assert bits_equal(uni.run(x4), fused[r0:r0 + nfs[4]])


def test_chain_mono_i16_odd_lengths(af, po):
    """Mono i16 clips of an ODD frame count (round-3 advisor: the fused launch stages mono i16 as one dword per frame
    pair and the buffer's range check works per dword, so the last sample of an odd clip was either lost or joined by
    the 2 bytes behind the clip).  Lengths chosen so that the last STFT frame of the 16 kHz signal ends on the clip's
    final samples: fused == stage-by-stage bit for bit, both within tolerance of the oracle, for a single odd clip, a
    ragged batch whose last clip is odd, and a uniform batch of odd clips (which leaves the fused launch)."""
    taps = lowpass_taps()
    # 16 kHz length 1024 + 256 k  <=>  48 kHz length 3 * that; -1 / -2 / +1 make it odd around the frame boundary
    for n in (3 * (1024 + 256 * 40) + 1, 3 * (1024 + 256 * 17) - 1, 3 * (1024 + 256 * 5) + 3):
        assert n & 1
        x = _as_i16(synth_clip(910 + n % 7, n, 1, 48000))
        chain = af.ChainPipeline(input_rate=48000, channels=1, output_rate=16000, fir_taps=taps, sample_format=1)
        chain.set_batch_uniform(1, n)
        assert chain.n_launches() == 2
        fused = chain.run(x)
        chain.set_option("unfused", 1)
        staged = chain.run(x)
        chain.set_option("unfused", -1)
        assert bits_equal(fused, staged), n
        xf = x.astype(np.float32) / np.float32(32768.0)
        assert frame_rel_err(fused, oracle_chain(po, xf, 1, 48000, 16000, taps)) < 2e-5
        # behind the clip sits a loud sample that must not leak into its last frame
        pad = np.concatenate([x, np.array([32767], np.int16)])
        chain.set_batch(np.array([0, n], np.int64))
        assert bits_equal(chain.run(pad), fused), n
    # ragged: even clips, the last one odd (its base stays even, the fused launch applies)
    lens = [40000, 11520 * 2, 3 * (1024 + 256 * 9) + 1]
    clips = [_as_i16(synth_clip(920 + i, m, 1, 48000)) for i, m in enumerate(lens)]
    chain = af.ChainPipeline(input_rate=48000, channels=1, output_rate=16000, fir_taps=taps, sample_format=1)
    chain.set_batch(np.concatenate([[0], np.cumsum(lens)]))
    assert chain.n_launches() == 2
    fused = chain.run(np.concatenate(clips))
    chain.set_option("unfused", 1)
    assert bits_equal(chain.run(np.concatenate(clips)), fused)
    chain.set_option("unfused", -1)
    row = 0
    for xc, m in zip(clips, lens):
        nf = chain.n_frames(m)
        assert frame_rel_err(fused[row:row + nf], oracle_chain(po, xc.astype(np.float32) / np.float32(32768.0), 1, 48000, 16000, taps)) < 2e-5
        row += nf
    # uniform batch of odd clips: every second base is odd -> stage by stage, same rows as each clip alone
    n = 3 * (1024 + 256 * 7) + 1
    clips = [_as_i16(synth_clip(930 + i, n, 1, 48000)) for i in range(3)]
    uni = af.ChainPipeline(input_rate=48000, channels=1, output_rate=16000, fir_taps=taps, sample_format=1)
    uni.set_batch_uniform(3, n)
    assert uni.n_launches() > 2
    got = uni.run(np.concatenate(clips))
    one = af.ChainPipeline(input_rate=48000, channels=1, output_rate=16000, fir_taps=taps, sample_format=1)
    one.set_batch_uniform(1, n)
    nf = one.n_frames(n)
    for i, xc in enumerate(clips):
        assert bits_equal(got[i * nf:(i + 1) * nf], one.run(xc))


@pytest.mark.parametrize("fi,fo,ch,fmt,with_fir", [(44100, 16000, 2, 0, True), (44100, 16000, 1, 1, True),
                                                   (48000, 44100, 2, 1, True), (44100, 16000, 2, 0, False),
                                                   (16000, 16000, 2, 0, True), (16000, 16000, 1, 0, False),
                                                   (22050, 16000, 1, 0, True)])
def test_chain_ragged_without_a_fused_plan(af, po, fi, fo, ch, fmt, with_fir):
    """Configurations without the fused launch (44.1 k input: ten block matrices; 48 -> 44.1 k: streamed matrices; no
    resampler; no FIR) on a ragged batch: the clips run as ONE strip with zero gaps -- a gather, one resampler launch,
    one FIR launch, the feature kernel, whatever the number of clips -- and every clip's rows are the rows of that clip
    alone through a uniform batch, bit for bit, and the oracle's within tolerance."""
    taps = lowpass_taps() if with_fir else None
    lens = [30000, 44100, 5000, 1, 0, 70001, 2047]
    clips = [synth_clip(800 + i, n, ch, fi) for i, n in enumerate(lens)]
    if fmt:
        clips = [_as_i16(x) for x in clips]
    chain = af.ChainPipeline(input_rate=fi, channels=ch, output_rate=fo, fir_taps=taps, sample_format=fmt)
    chain.set_option("unfused", 1)              # (44.1 k -> 16 k with a FIR has a fused launch since round 5: the strip is what is tested here)
    chain.set_batch(np.concatenate([[0], np.cumsum(lens)]))
    resample = fi != fo
    assert chain.n_launches() == 2 + (2 if resample and with_fir else 1 if resample else 0) + (1 if with_fir else 0)
    mel = chain.run(np.concatenate([x.reshape(-1) for x in clips]))
    alone = af.ChainPipeline(input_rate=fi, channels=ch, output_rate=fo, fir_taps=taps, sample_format=fmt)
    row = 0
    for x, n in zip(clips, lens):
        nf = chain.n_frames(n)
        if nf:
            alone.set_batch_uniform(1, n)
            one = alone.run(x.reshape(-1))
            assert bits_equal(mel[row:row + nf], one)
            xf = x.astype(np.float32) / 32768.0 if fmt else x
            assert frame_rel_err(mel[row:row + nf], oracle_chain(po, xf, ch, fi, fo, taps)) < 2e-5
        row += nf
    assert row == mel.shape[0]


def test_chain_ragged_strip_many_clips(af):
    """600 ragged clips at 44.1 kHz stereo: still five launches, every clip bit-equal to itself alone."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(3)
    lens = [int(v) for v in rng.integers(2000, 90000, 600)]
    taps = lowpass_taps()
    gen = torch.Generator().manual_seed(9)
    x = (torch.rand(sum(lens), 2, generator=gen) * 2 - 1).mul_(0.4).numpy()
    chain = af.ChainPipeline(input_rate=44100, channels=2, output_rate=16000, fir_taps=taps)
    chain.set_option("unfused", 1)
    offs = np.concatenate([[0], np.cumsum(lens)])
    chain.set_batch(offs)
    assert chain.n_launches() == 5
    mel = chain.run(x.reshape(-1))
    alone = af.ChainPipeline(input_rate=44100, channels=2, output_rate=16000, fir_taps=taps)
    rows = np.concatenate([[0], np.cumsum([chain.n_frames(n) for n in lens])])
    for i in (0, 1, 299, 598, 599):
        alone.set_batch_uniform(1, lens[i])
        one = alone.run(x[offs[i]:offs[i + 1]].reshape(-1))
        assert bits_equal(mel[rows[i]:rows[i + 1]], one)
    assert rows[-1] == mel.shape[0]


def test_chain_pipelined_form_same_bits(af):
    """af_chain_set_option("pipeline", k): the two launches side by side on two streams, chunk by chunk (measured slower than
    one after the other on this chip, hence opt-in) -- same rows bit for bit, also when the chunks do not divide the batch."""
    taps = lowpass_taps()
    n_clips, n = 37, 30011
    x = np.concatenate([synth_clip(1500 + i, n, 2, 48000) for i in range(n_clips)])
    chain = af.ChainPipeline(fir_taps=taps)
    chain.set_batch_uniform(n_clips, n)
    ref = chain.run(x)
    assert chain.n_launches() == 2
    for k in (1, 3, 8, 37, 64):
        chain.set_option("pipeline", k)
        assert chain.n_launches() == 2 * min(k, n_clips)
        assert bits_equal(chain.run(x), ref)
    chain.set_option("pipeline", 0)
    assert chain.n_launches() == 2 and bits_equal(chain.run(x), ref)


def test_chain_ragged_random_configurations(af, po):
    """Seeded sweep over rate pairs (fused, ten-wave, streamed, VALU-fallback ratios, no resampler), FIR lengths up to
    several banded passes, formats and clip lengths: every clip of the ragged batch equals itself alone, bit for bit, and
    the first one the oracle within tolerance."""
    rng = np.random.default_rng(2024)
    rates = [(48000, 16000), (44100, 16000), (32000, 16000), (48000, 44100), (22050, 16000), (11025, 16000), (8000, 16000),
             (16000, 16000), (44100, 48000), (96000, 16000), (44100, 8000)]
    for case in range(22):
        fi, fo = rates[int(rng.integers(len(rates)))]
        ch, fmt = int(rng.integers(1, 3)), int(rng.integers(0, 2))
        K = int(rng.choice([0, 1, 17, 64, 129, 130, 177, 178, 400, 1000]))
        taps = lowpass_taps(K, fc=0.4 * min(fo, fi), sr=float(fo)) if K else None
        n_clips = int(rng.integers(1, 7))
        lens = [int(v) for v in rng.integers(0, 60000, n_clips)]
        clips = [synth_clip(1000 + 10 * case + i, n, ch, fi) for i, n in enumerate(lens)]
        if fmt:
            clips = [_as_i16(x) for x in clips]
        chain = af.ChainPipeline(input_rate=fi, channels=ch, output_rate=fo, fir_taps=taps, sample_format=fmt)
        total = chain.set_batch(np.concatenate([[0], np.cumsum(lens)]))
        mel = chain.run(np.concatenate([x.reshape(-1) for x in clips]) if sum(lens) else np.zeros(0, clips[0].dtype))
        assert mel.shape == (total, 80), (case, fi, fo, ch, fmt, K, lens)
        alone = af.ChainPipeline(input_rate=fi, channels=ch, output_rate=fo, fir_taps=taps, sample_format=fmt)
        row, checked = 0, False
        for x, n in zip(clips, lens):
            nf = chain.n_frames(n)
            if nf:
                alone.set_batch_uniform(1, n)
                one = alone.run(x.reshape(-1))
                assert bits_equal(mel[row:row + nf], one), (case, fi, fo, ch, fmt, K, lens)
                if not checked:
                    xf = x.astype(np.float32) / 32768.0 if fmt else x
                    assert frame_rel_err(one, oracle_chain(po, xf, ch, fi, fo, taps)) < 3e-5, (case, fi, fo, ch, fmt, K)
                    checked = True
            row += nf
        assert row == total


def test_chain_ragged_strip_more_clips_than_a_grid_dimension(af):
    """66 000 short clips (the gather's grid.y is 16 bits: two launches of it), most of them too short for a row."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(5)
    lens = rng.integers(50, 400, 66000)
    lens[[7, 40000, 65999]] = [9000, 12000, 7000]                     # three clips long enough for rows
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    gen = torch.Generator().manual_seed(10)
    x = (torch.rand(int(offs[-1]), 2, generator=gen) * 2 - 1).mul_(0.4).numpy()
    taps = lowpass_taps()
    chain = af.ChainPipeline(input_rate=44100, channels=2, output_rate=16000, fir_taps=taps)
    chain.set_option("unfused", 1)
    total = chain.set_batch(offs)
    assert chain.n_launches() == 6 and total == sum(chain.n_frames(int(n)) for n in lens[[7, 40000, 65999]])
    mel = chain.run(x.reshape(-1))
    alone = af.ChainPipeline(input_rate=44100, channels=2, output_rate=16000, fir_taps=taps)
    row = 0
    for i in (7, 40000, 65999):
        alone.set_batch_uniform(1, int(lens[i]))
        one = alone.run(x[offs[i]:offs[i + 1]].reshape(-1))
        assert bits_equal(mel[row:row + one.shape[0]], one)
        row += one.shape[0]
    assert row == mel.shape[0]


def test_chain_ragged_strip_in_groups_beyond_2_30_samples(af):
    """1.2e9 mono frames in 240 clips: the strip is cut into two groups at a clip start (the banded kernels address a
    launch with 32-bit offsets); clips on both sides of the cut equal themselves alone."""
    torch = pytest.importorskip("torch")
    lens = np.full(240, 5_000_000, dtype=np.int64)
    lens[::7] += 12345
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(11)
    pcm = (torch.rand(int(offs[-1]), device=dev, generator=gen) * 2 - 1) * 0.4
    taps = lowpass_taps()
    chain = af.ChainPipeline(input_rate=44100, channels=1, output_rate=16000, fir_taps=taps)
    chain.set_option("unfused", 1)
    total = chain.set_batch(offs)
    assert chain.n_launches() == 1 + 1 + 2 + 1 + 2                    # features, gather, 2 x resampler, gap zeroing, 2 x FIR
    mel = torch.empty(total, 80, device=dev)
    chain.run_device(pcm.data_ptr(), mel.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    alone = af.ChainPipeline(input_rate=44100, channels=1, output_rate=16000, fir_taps=taps)
    rows = np.concatenate([[0], np.cumsum([chain.n_frames(int(n)) for n in lens])])
    for i in (0, 213, 214, 215, 239):                                 # (the cut falls behind clip 213 or 214)
        alone.set_batch_uniform(1, int(lens[i]))
        one = alone.run(pcm[offs[i]:offs[i + 1]].cpu().numpy())
        assert bits_equal(mel[rows[i]:rows[i + 1]].cpu().numpy(), one)


# ---- ChainStream (af_chain_stream_*): config 5 on the whole chain ----------------------------------------------
@pytest.mark.parametrize("ch,fmt,n", [(2, 0, 184320 * 3 + 50001), (1, 1, 184320 * 2 + 7), (2, 0, 190000),
                                      (2, 1, 184320 + 2200), (1, 0, 2000), (2, 0, 0)])
def test_chain_stream_small_steps_equal_one_shot(af, po, ch, fmt, n):
    """Ragged pushes over several steps (the smallest step: 184 320 input frames = 61 440 FIR outputs = 16 tiles of
    the fused launch = 15 units of the feature kernel), a partial last step, streams shorter than one step, i16 and
    mono: rows of pushes + flush == ChainPipeline.run on the whole stream, bit for bit."""
    taps = lowpass_taps()
    x = synth_clip(900 + ch + fmt, n, ch, 48000)
    if fmt:
        x = _as_i16(x)
    st = af.ChainStream(step_frames=1, channels=ch, fir_taps=taps, sample_format=fmt)
    assert st.step_frames == 184320 and st.rows_per_step == 240
    rng = np.random.default_rng(n % 97)
    rows, pos = [], 0
    while pos < n:
        k = int(rng.integers(0, 120000))
        rows.append(st.push(x[pos * ch:(pos + k) * ch]))
        pos += k
    rows.append(st.flush())
    got = np.concatenate(rows)
    one = af.ChainPipeline(channels=ch, fir_taps=taps, sample_format=fmt)
    one.set_batch_uniform(1, n)
    assert got.shape[0] == st.total_rows(n) == one.n_frames(n)
    if n:
        full = one.run(x)
        assert bits_equal(got, full)
        if got.shape[0]:
            xf = x.astype(np.float32) / np.float32(32768.0) if fmt else x
            assert frame_rel_err(got, oracle_chain(po, xf, ch, 48000, 16000, taps)) < 2e-5
    with pytest.raises(af.AudioError):
        st.push(x[:8 * ch] if n else np.zeros(8 * ch, x.dtype))                  # pushed after flush
    st.reset()
    assert st.push(np.zeros(100 * ch, x.dtype)).shape[0] == 0


def test_chain_stream_ten_minutes_bit_equal_to_one_shot(af):
    """>= 10 min of 48 kHz stereo pushed in ragged pieces through af_chain_stream_* (resampler history, FIR history
    and STFT carry owned by the handle; one hipGraph replay per ~19 s step) == af_chain_run on the whole stream,
    bit for bit; per-step latency printed."""
    torch = pytest.importorskip("torch")
    sr, ch = 48000, 2
    n = 10 * 60 * sr + 54321
    gen = torch.Generator().manual_seed(12)
    x = (torch.rand(n, ch, generator=gen) * 2 - 1).mul_(0.4).numpy()
    taps = lowpass_taps()
    st = af.ChainStream(step_frames=1 << 20, fir_taps=taps)
    assert st.step_frames == 6 * 184320                       # the multiple of the step quantum nearest to 2^20
    rng = np.random.default_rng(8)
    rows, dev_ms, wall_ms, pos = [], [], [], 0
    while pos < n:
        k = int(rng.integers(1, 700000))
        r = st.push(x[pos:pos + k])
        if r.shape[0] and st.stats()["steps"] > 1:
            s = st.stats()
            dev_ms.append(s["device_ms"]); wall_ms.append(s["wall_ms"])
        rows.append(r)
        pos += k
    rows.append(st.flush())
    got = np.concatenate(rows)
    one = af.ChainPipeline(fir_taps=taps)
    one.set_batch_uniform(1, n)
    full = one.run(x.reshape(-1))
    need0 = st.step_frames + 3 * 767 + 33                     # step 0 also needs the STFT halo and the resampler's look-ahead
    assert got.shape == full.shape == (st.total_rows(n), 80) and st.stats()["steps"] == 1 + (n - need0) // st.step_frames == 26
    assert bits_equal(got, full)
    step_s = st.step_frames / sr
    print(f"\nchain stream: {len(dev_ms)} replayed steps of {step_s:.1f} s audio; device ms/step median "
          f"{np.median(dev_ms):.2f} max {np.max(dev_ms):.2f}; wall ms/step median {np.median(wall_ms):.2f}; "
          f"real-time factor {step_s / (np.median(wall_ms) * 1e-3):.0f}x")
    assert np.median(wall_ms) * 1e-3 < step_s / 100


@pytest.mark.parametrize("ch,fmt,world", [(2, 0, 8), (2, 1, 3), (1, 0, 5), (1, 1, 2)])
def test_chain_stream_time_sharded_equals_one_shot(af, ch, fmt, world):
    """Config 5 on the chain across GPUs: the recording's steps dealt to `world` ranks (shard.chain_stream_shards), each
    rank seeks to its first step, pushes ITS frames (history first) and the last one flushes; the ranks' rows in rank
    order == af_chain_run on the whole recording, bit for bit.  The ranks run one after the other on this one GPU; the
    world_size-2 gloo test covers the dealing across processes."""
    torch = pytest.importorskip("torch")
    from audio_flow_rs_amd.shard import chain_stream_shards
    sr = 48000
    taps = lowpass_taps()
    st = af.ChainStream(step_frames=184320, channels=ch, fir_taps=taps, sample_format=fmt)
    n = st.first_step_frames + 10 * st.step_frames + 70001         # 11 whole steps and a tail
    gen = torch.Generator().manual_seed(100 + world)
    x = (torch.rand(n, ch, generator=gen) * 2 - 1).mul_(0.4).numpy()
    if fmt:
        x = _as_i16(x)
    one = af.ChainPipeline(channels=ch, fir_taps=taps, sample_format=fmt)
    one.set_batch_uniform(1, n)
    full = one.run(x.reshape(-1))
    shards = chain_stream_shards(n, world, st.first_step_frames, st.step_frames, st.history_frames)
    assert [s[0] for s in shards][0] == 0 and shards[-1][1] == 11 and shards[-1][4]
    rows = []
    for k0, k1, lo, hi, flush in shards:
        if k0 == k1 and not flush:
            continue
        assert st.seek(k0) == lo
        r = st.push(x[lo:hi])
        rps = st.step_frames // 3 // 256                          # rows per step: 48 k -> 16 k, hop 256
        assert r.shape[0] == (k1 - k0) * rps
        assert bits_equal(r, full[k0 * rps:k1 * rps])
        rows.append(r)
        if flush:
            rows.append(st.flush())
    got = np.concatenate(rows)
    assert got.shape == full.shape
    assert bits_equal(got, full)
    # a rank that is handed too few frames for its history cannot flush
    st.seek(3)
    st.push(x[:100])
    with pytest.raises(af.AudioError, match="history"):
        st.flush()
    # seek(0) is a plain restart
    assert st.seek(0) == 0
    r0 = st.push(x[:st.first_step_frames])
    assert bits_equal(r0, full[:r0.shape[0]])


# ---- 44.1 kHz input: the ten-matrix fused launch (resfir10_kernel) and the chain stream on top of it ------------------
@pytest.mark.parametrize("ch,fmt", [(2, 0), (1, 0), (2, 1), (1, 1)])
@pytest.mark.parametrize("K", [129, 100, 65])
def test_chain_44k1_fused_equals_separate_launches(af, po, ch, fmt, K):
    """44.1 k -> 16 k (capture.rs:60-80 takes whatever rate the device offers; BASELINE config 3's ratio): downmix +
    resampler + FIR as ONE launch -- ten block matrices, the K - 1 resampled samples in front of a tile carried in LDS,
    a warm pass where a workgroup's range starts inside a clip -- against the stand-alone kernels, bit for bit (FIRs of
    34-65 and 98-129 taps share the stand-alone chains' step layout); uniform and ragged batches, clips that end on /
    one over / one under a tile edge (5 120 outputs = 14 112 input frames), a clip shorter than the FIR history."""
    taps = lowpass_taps()[:K].copy()
    for lens in ([44100 * 2 + 17] * 3, [14112 * 2] * 2, [14112 * 2 + 3, 14112 * 2 - 3, 5, 0, 70001, 300, 1, 14113 * 4]):
        if ch == 1 and fmt == 1:
            lens = [n + (n & 1) for n in lens]                 # mono i16 clips start on even frames
        clips = [synth_clip(700 + i, n, ch, 44100) for i, n in enumerate(lens)]
        if fmt:
            clips = [_as_i16(x) for x in clips]
        x = np.concatenate([c.reshape(-1) for c in clips])
        rows = []
        for unfused in (0, 1):
            c = af.ChainPipeline(input_rate=44100, channels=ch, output_rate=16000, fir_taps=taps, sample_format=fmt)
            c.set_option("unfused", unfused)
            if len(set(lens)) == 1:
                c.set_batch_uniform(len(lens), lens[0])
            else:
                c.set_batch(np.concatenate([[0], np.cumsum(lens)]))
            assert c.n_launches() == 2 if not unfused else c.n_launches() >= 3
            rows.append(c.run(x))
        assert bits_equal(rows[0], rows[1])
    # the last batch against the float64 oracle, clip by clip
    row = 0
    for cx, n in zip(clips, lens):
        nf = c.n_frames(n)
        if nf:
            xf = cx.astype(np.float32) / np.float32(32768.0) if fmt else cx
            assert frame_rel_err(rows[0][row:row + nf], oracle_chain(po, xf, ch, 44100, 16000, taps)) < 2e-5
        row += nf


def test_chain_44k1_full_size_fused_equals_stages(af):
    """256 x 30 s stereo 44.1 kHz (2.7 GB of PCM): every workgroup of the fused launch owns one clip's 94 tiles; rows equal
    the three-launch form, identical clips give identical rows."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    n_clips, n = 256, 44100 * 30
    x = (torch.rand(n_clips, n, 2, device=dev) * 2 - 1) * 0.5
    x[200] = x[3]
    out = []
    for unfused in (0, 1):
        c = af.ChainPipeline(input_rate=44100, channels=2, output_rate=16000, fir_taps=lowpass_taps())
        c.set_option("unfused", unfused)
        total = c.set_batch_uniform(n_clips, n)
        mel = torch.empty(total, 80, device=dev)
        c.run_device(x.data_ptr(), mel.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        out.append(mel)
    assert bits_equal(out[0], out[1]) and bool(torch.isfinite(out[0]).all())
    m = out[0].view(n_clips, -1, 80)
    assert bits_equal(m[200], m[3]) and not bits_equal(m[200], m[4])


@pytest.mark.parametrize("ch,fmt,n", [(2, 0, 56448 * 9 + 50001), (1, 0, 56448 * 5 + 7), (2, 1, 56448 * 4 + 2200), (1, 1, 56448 * 6 + 12),
                                      (2, 0, 60000), (1, 0, 2000)])
def test_chain_stream_44k1_equals_one_shot(af, po, ch, fmt, n):
    """af_chain_stream_* at 44.1 kHz (the carry / flush contract of BatchResampler, resampler.rs:132-166, on the whole
    chain): ragged pushes over several steps (the smallest step: 56 448 input frames = 20 480 FIR outputs = 4 tiles of
    the ten-matrix launch = 5 units of the feature kernel; every replayed step begins with the warm pass of the tile in
    front of it), a partial last step, streams shorter than a step: rows of pushes + flush == ChainPipeline.run on the
    whole stream, bit for bit."""
    taps = lowpass_taps()
    x = synth_clip(950 + ch + fmt, n, ch, 44100)
    if fmt:
        x = _as_i16(x)
    st = af.ChainStream(step_frames=1, input_rate=44100, channels=ch, fir_taps=taps, sample_format=fmt)
    assert st.step_frames == 56448 and st.rows_per_step == 80
    rng = np.random.default_rng(n % 89)
    rows, pos = [], 0
    while pos < n:
        k = int(rng.integers(0, 90000))
        rows.append(st.push(x[pos * ch:(pos + k) * ch]))
        pos += k
    rows.append(st.flush())
    got = np.concatenate(rows)
    one = af.ChainPipeline(input_rate=44100, channels=ch, fir_taps=taps, sample_format=fmt)
    one.set_batch_uniform(1, n)
    assert one.n_launches() == 2
    assert got.shape[0] == st.total_rows(n) == one.n_frames(n)
    full = one.run(x)
    assert bits_equal(got, full)
    if got.shape[0]:
        xf = x.astype(np.float32) / np.float32(32768.0) if fmt else x
        assert frame_rel_err(got, oracle_chain(po, xf, ch, 44100, 16000, taps)) < 2e-5


def test_chain_stream_44k1_time_sharded_equals_one_shot(af):
    """the steps of a 44.1 kHz recording dealt to 3 ranks (seek: history first, then the rank's own steps): rows in rank
    order == af_chain_run on the whole recording, bit for bit"""
    torch = pytest.importorskip("torch")
    from audio_flow_rs_amd.shard import chain_stream_shards
    taps = lowpass_taps()
    st = af.ChainStream(step_frames=56448 * 3, input_rate=44100, channels=2, fir_taps=taps)
    n = st.first_step_frames + 7 * st.step_frames + 40001
    gen = torch.Generator().manual_seed(44)
    x = (torch.rand(n, 2, generator=gen) * 2 - 1).mul_(0.4).numpy()
    one = af.ChainPipeline(input_rate=44100, channels=2, fir_taps=taps)
    one.set_batch_uniform(1, n)
    full = one.run(x.reshape(-1))
    rows = []
    for k0, k1, lo, hi, flush in chain_stream_shards(n, 3, st.first_step_frames, st.step_frames, st.history_frames):
        if k0 == k1 and not flush:
            continue
        assert st.seek(k0) == lo
        rows.append(st.push(x[lo:hi]))
        if flush:
            rows.append(st.flush())
    got = np.concatenate(rows)
    assert got.shape == full.shape and bits_equal(got, full)


def test_chain_stream_needs_the_fused_launch(af):
    with pytest.raises(af.AudioError, match="no fused"):
        af.ChainStream(input_rate=48000, output_rate=44100, fir_taps=lowpass_taps())      # 147 streamed block matrices
    with pytest.raises(af.AudioError, match="no fused"):
        af.ChainStream()                                          # no FIR


def test_features_per_handle_option(af):
    """af_features_set_option("kernel", 1) routes ONE handle to the general kernel; another handle is untouched."""
    x = synth_clip(77, 30000, 2, 48000)
    a, b = af.FeaturePipeline(), af.FeaturePipeline()
    a.set_batch_uniform(1, 30000); b.set_batch_uniform(1, 30000)
    base = a.run(x)
    lib = af.library()
    assert lib.af_features_set_option(a._h, b"kernel", 1) == 0
    gen = a.run(x)
    assert bits_equal(b.run(x), base)                        # b still on the hop-256 kernel
    assert frame_rel_err(gen, base) < 2e-6 and lib.af_features_set_option(a._h, b"bogus", 1) != 0


@pytest.mark.parametrize("ch,fmt", [(2, 0), (1, 1)])
def test_chain_with_speech_frames(af, po, ch, fmt):
    """48 kHz capture -> to_mono -> 16 kHz -> FIR -> 25 ms / 10 ms frames (n_fft 400, hop 160: `af_chain_set_option(c, "n_fft", 400)`,
    the STFT / mel stage then runs the matrix-product DFT kernel) -> 80 mel, against the chained oracle; ragged clips."""
    taps = lowpass_taps()
    lens = [48000, 30001, 7000]
    clips = [synth_clip(190 + i, n, ch, 48000) for i, n in enumerate(lens)]
    if fmt:
        clips = [np.round(c * 32767.0).astype(np.int16) for c in clips]
    chain = af.ChainPipeline(input_rate=48000, channels=ch, output_rate=16000, fir_taps=taps, hop=160, sample_format=fmt)
    chain.set_option("n_fft", 400)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    total = chain.set_batch(off)
    got = chain.run(np.concatenate(clips))
    assert got.shape == (total, 80)
    row = 0
    for c in clips:
        x = c.astype(np.float32) / np.float32(32768.0) if fmt else c
        mono = po.to_mono(x, ch)
        y = po.fir(po.resample(mono, 48000, 16000).astype(np.float32), taps).astype(np.float32)
        ref = po.features(y, 1, 16000.0, n_fft=400, hop=160)
        assert frame_rel_err(got[row:row + ref.shape[0]], ref) < 1e-5
        row += ref.shape[0]
    assert row == total
