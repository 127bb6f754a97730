"""Frame lengths outside {1024, 512, 256} (`features_dft.hip`: the DFT as a matrix product on the f32 matrix cores;
400 samples = 25 ms at 16 kHz is the usual speech frame, websocket.rs:274 names the 16 kHz format): the same contract
as every other configuration -- frame counts and placement exact, rows within 1e-5 of the float64 oracle per frame
(docs/OPS.md "Tolerance"), through the C ABI."""
import numpy as np
import pytest

from util import bits_equal, frame_rel_err, synth_clip

pytestmark = pytest.mark.gpu
TOL = 1e-5

CASES = [
    dict(sample_rate=16000, channels=1, n_fft=400, hop=160),                      # 25 ms / 10 ms, the speech front end
    dict(sample_rate=16000, channels=2, n_fft=400, hop=160, n_mels=40),
    dict(sample_rate=48000, channels=2, n_fft=480, hop=240),                      # 10 ms frames at 48 kHz
    dict(sample_rate=16000, channels=1, n_fft=400, hop=160, sample_format=1),     # i16 (odd clip starts: 2-byte aligned only)
    dict(sample_rate=16000, channels=2, n_fft=400, hop=160, sample_format=1),
    dict(sample_rate=16000, channels=2, n_fft=320, hop=160, channel_mode=1),      # separate channels
    dict(sample_rate=16000, channels=1, n_fft=400, hop=160, log_mel=True),
    dict(sample_rate=16000, channels=1, n_fft=400, hop=160, mel_scale=1, mel_norm=1, n_mels=128),
    dict(sample_rate=8000, channels=1, n_fft=200, hop=80, n_mels=23),             # n_mels not a multiple of 4
    dict(sample_rate=44100, channels=1, n_fft=1000, hop=512),                     # two waves per workgroup
    dict(sample_rate=44100, channels=1, n_fft=600, hop=1024),                     # hop > n_fft, one wave per workgroup
    dict(sample_rate=16000, channels=1, n_fft=100, hop=64),
    dict(sample_rate=16000, channels=1, n_fft=33, hop=64),                        # odd frame length, K padded to 48
    dict(sample_rate=16000, channels=3, n_fft=400, hop=160),                      # to_mono first (capture.rs:30-42)
    dict(sample_rate=22050, channels=3, n_fft=36, hop=128),                       # rows padded to four K groups (64 entries for 18 pairs):
    dict(sample_rate=22050, channels=1, n_fft=36, hop=128),                       # the first version's 16-byte row loads left the frame here
    dict(sample_rate=22050, channels=2, n_fft=40, hop=64),
]


def oracle_rows(po, clip, kw):
    ch, sr, n_fft, hop = kw["channels"], kw["sample_rate"], kw["n_fft"], kw["hop"]
    x = clip.astype(np.float32) / np.float32(32768.0) if kw.get("sample_format") else clip
    W = po.mel_weights(float(sr), n_fft, kw.get("n_mels", 80), scale=kw.get("mel_scale", 0), norm=kw.get("mel_norm", 0))
    if kw.get("channel_mode"):
        xs = x.reshape(-1, ch)
        return np.concatenate([po.mel_apply(po.stft_power(np.ascontiguousarray(xs[:, c]), n_fft, hop), W) for c in range(ch)])
    return po.mel_apply(po.stft_power(po.to_mono(x, ch), n_fft, hop), W)


@pytest.mark.parametrize("kw", CASES, ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()))
def test_any_frame_length_against_the_oracle(af, po, kw):
    """Ragged batch: a long clip, one that ends inside a 16-frame unit, exactly one frame, one shorter than a frame (no
    rows), one more unit-and-a-bit.  Rows of every clip against the float64 oracle; the frame counts are the reference's
    `1 + (n - n_fft) / hop` rule."""
    ch, sr, n_fft, hop = kw["channels"], kw["sample_rate"], kw["n_fft"], kw["hop"]
    lens = [n_fft + hop * 50 + 3, n_fft + hop * 16 - 1, n_fft, n_fft - 1, n_fft + hop * 17 + hop // 2, 7]
    clips = [synth_clip(700 + i, n, ch, sr) for i, n in enumerate(lens)]
    if kw.get("sample_format"):
        clips = [np.round(c * 32767.0).astype(np.int16) for c in clips]
    pipe = af.FeaturePipeline(**kw)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    total = pipe.set_batch(off)
    sep = 2 if kw.get("channel_mode") else 1
    want = [sep * (1 + (n - n_fft) // hop if n >= n_fft else 0) for n in lens]
    assert total == sum(want)
    got = pipe.run(np.concatenate(clips))
    assert got.shape == (total, kw.get("n_mels", 80)) and np.isfinite(got).all()
    row = 0
    for c, nrows in zip(clips, want):
        if nrows:
            ref = oracle_rows(po, c, kw)
            assert ref.shape[0] == nrows
            g = got[row:row + nrows]
            if kw.get("log_mel"):
                assert frame_rel_err(10.0 ** g.astype(np.float64), np.maximum(ref, 1e-10)) < TOL   # judged on the mel values (per-frame norm)
            else:
                assert frame_rel_err(g, ref) < TOL
        row += nrows


def test_power_rows_and_the_stft_handle(af, po):
    """`Stft(n_fft=400, hop=160)`: rows [frame][201] against the oracle's DFT sum; the same rows from the feature handle
    with power output, bit for bit; `MelFilterbank` on those rows == the fused mel rows within float32 summation order."""
    x = synth_clip(41, 16000, 1, 16000)
    st = af.Stft(n_fft=400, hop=160)
    assert st.n_bins() == 201
    p = st.power(x)
    ref = po.stft_power(x, 400, 160)
    assert p.shape == ref.shape == (98, 201)
    assert frame_rel_err(p, ref) < TOL
    fp = af.FeaturePipeline(sample_rate=16000, channels=1, n_fft=400, hop=160, output=af._ffi.AF_OUT_POWER)
    fp.set_batch_uniform(1, 16000)
    assert bits_equal(fp.run(x), p)
    mel = af.MelFilterbank(sample_rate=16000, n_fft=400, n_mels=80)
    assert np.array_equal(mel.weights(), po.mel_weights(16000.0, 400, 80).astype(np.float32))
    fused = af.FeaturePipeline(sample_rate=16000, channels=1, n_fft=400, hop=160)
    fused.set_batch_uniform(1, 16000)
    assert frame_rel_err(mel.apply(p), np.asarray(fused.run(x), np.float64)) < 2e-6


def test_rows_do_not_depend_on_the_batch_or_the_position(af):
    """A clip's rows are the same bits alone, in a batch behind clips of other lengths (any unit / wave / workgroup it
    lands on), and at an offset that is only 4-byte aligned."""
    kw = dict(sample_rate=16000, channels=1, n_fft=400, hop=160)
    x = synth_clip(55, 16000 * 3 + 77, 1, 16000)
    alone = af.FeaturePipeline(**kw)
    alone.set_batch_uniform(1, x.size)
    a = alone.run(x)
    lens = [401, 5000, 399, x.size, 12345]
    clips = [synth_clip(60 + i, n, 1, 16000) for i, n in enumerate(lens)]
    clips[3] = x
    pipe = af.FeaturePipeline(**kw)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    pipe.set_batch(off)
    got = pipe.run(np.concatenate(clips))
    r0 = pipe.n_frames(401) + pipe.n_frames(5000)
    assert bits_equal(got[r0:r0 + a.shape[0]], a)


def test_stream_and_one_shot_agree(af):
    """`FeatureStream` over a 400 / 160 stream in steps of 16 * hop * k frames == one-shot framing of the whole signal,
    bit for bit (the carry is n_fft - hop samples, resampler.rs:132-166 is the reference's carry / flush analogue)."""
    kw = dict(sample_rate=16000, channels=1, n_fft=400, hop=160)
    x = synth_clip(77, 16000 * 4 + 123, 1, 16000)
    one = af.FeaturePipeline(**kw)
    one.set_batch_uniform(1, x.size)
    ref = one.run(x)
    s = af.FeatureStream(step_frames=16 * 160 * 4, **kw)
    rows = []
    for i in range(0, x.size, 3001):
        rows.append(s.push(x[i:i + 3001]))
    rows.append(s.flush())
    got = np.concatenate([r for r in rows if r.size])
    assert bits_equal(got, ref)


def test_whisper_shaped_batch_at_size(af, po):
    """256 clips x 30 s mono 16 kHz, n_fft 400 / hop 160 / 80 mels (2998 frames per clip): three clips against the
    oracle, every clip against clip 0's rows where the PCM repeats (placement), all rows finite."""
    import torch
    dev = torch.device("cuda:0")
    n, clips = 480000, 256
    base = [synth_clip(900 + i, n, 1, 16000) for i in range(3)]
    x = torch.from_numpy(np.stack([base[i % 3] for i in range(clips)])).to(dev)
    pipe = af.FeaturePipeline(sample_rate=16000, channels=1, n_fft=400, hop=160)
    total = pipe.set_batch_uniform(clips, n)
    nf = pipe.n_frames(n)
    assert nf == 2998 and total == clips * nf
    mel = torch.empty(total, 80, device=dev)
    pipe.run_device(x.data_ptr(), mel.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    m = mel.cpu().numpy().reshape(clips, nf, 80)
    assert np.isfinite(m).all()
    for i in range(3):
        assert frame_rel_err(m[i], po.features(base[i], 1, 16000.0, n_fft=400, hop=160)) < TOL
    for c in range(3, clips):
        assert bits_equal(m[c], m[c % 3])


def test_one_and_two_waves_per_unit_agree(af, po):
    """`af_features_set_option(h, "dft_waves", 1)`: one wave multiplies every bin tile of a unit instead of two waves taking
    every other one (their partial mel sums meet in LDS): same rows within float32 summation order, both inside the tolerance;
    power rows (no sums across waves) bit for bit."""
    lib = af.library()
    kw = dict(sample_rate=16000, channels=1, n_fft=400, hop=160)
    x = synth_clip(33, 16000 * 2 + 55, 1, 16000)
    ref = po.features(x, 1, 16000.0, n_fft=400, hop=160)
    rows = []
    for waves in (2, 1):
        p = af.FeaturePipeline(**kw)
        assert lib.af_features_set_option(p._h, b"dft_waves", waves) == 0
        p.set_batch_uniform(1, x.size)
        rows.append(p.run(x))
        assert frame_rel_err(rows[-1], ref) < TOL
    assert frame_rel_err(rows[0], np.asarray(rows[1], np.float64)) < 2e-6
    pw = []
    for waves in (2, 1):
        p = af.FeaturePipeline(output=af._ffi.AF_OUT_POWER, **kw)
        assert lib.af_features_set_option(p._h, b"dft_waves", waves) == 0
        p.set_batch_uniform(1, x.size)
        pw.append(p.run(x))
    assert bits_equal(pw[0], pw[1])
    # the option belongs to handles this kernel serves
    assert lib.af_features_set_option(af.FeaturePipeline()._h, b"dft_waves", 1) == af._ffi.AF_ERR_INVALID_ARG
