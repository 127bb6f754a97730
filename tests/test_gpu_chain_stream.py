"""GPU parity of the chained path (BASELINE config 4: to_mono -> 48->16 kHz polyphase ->
FIR -> STFT -> mel) and of chunked streaming with a carried halo and a hipGraph-captured
step (BASELINE config 5).  Streaming == one-shot is bit-exact; the chain is compared with
the float64 oracle chain at the docs/OPS.md tolerance."""
import numpy as np
import pytest

from util import frame_rel_err, synth_clip

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
    assert np.array_equal(chunked, full)                         # bit-exact
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
    assert np.array_equal(got, full)                             # same frame indices, same bits


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
