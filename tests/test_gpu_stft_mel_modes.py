"""GPU parity of the other output modes of the fused path, through the C ABI: the power
spectrogram (BASELINE config 1), separate channels, log-mel, the
Slaney filterbank, and the stand-alone MelFilterbank contraction.  Tolerances as in
test_gpu_features.py: per frame ||got - ref||_2 / ||ref||_2 <= 1e-5 against float64.
"""
import numpy as np
import pytest

from util import bits_equal, frame_rel_err, synth_clip

pytestmark = pytest.mark.gpu
TOL = 1e-5


def test_config1_power_spectrogram(af, po):
    # BASELINE config 1: one 10 s mono clip at 16 kHz -> 622 frames x 513 bins
    x = synth_clip(1, 160000, 1, 16000)
    st = af.Stft()
    assert st.n_frames(160000) == 622 and st.n_bins() == 513
    P = st.power(x)
    assert P.shape == (622, 513)
    ref = po.stft_power(x)
    assert frame_rel_err(P, ref) < TOL


@pytest.mark.parametrize("ch,hop", [(2, 256), (1, 160), (2, 512)])
def test_power_output_ragged_batch(af, po, ch, hop):
    lens = [1024, 5000, 1023, 16 * hop + 1024, 40000]
    clips = [synth_clip(10 + i, n, ch, 48000) for i, n in enumerate(lens)]
    pipe = af.FeaturePipeline(sample_rate=48000, channels=ch, hop=hop, output=af._ffi.AF_OUT_POWER)
    offs = np.concatenate([[0], np.cumsum(lens)])
    total = pipe.set_batch(offs)
    assert pipe.row_len() == 513
    got = pipe.run(np.concatenate(clips))
    assert got.shape == (total, 513)
    row = 0
    for c in clips:
        ref = po.stft_power(po.to_mono(c, ch), hop=hop)
        assert frame_rel_err(got[row:row + ref.shape[0]], ref) < TOL
        row += ref.shape[0]
    assert row == total


def test_power_output_i16(af, po):
    x = (synth_clip(3, 30000, 2, 48000) * 32767).astype(np.int16)
    pipe = af.FeaturePipeline(channels=2, sample_format=af._ffi.AF_PCM_I16, output=af._ffi.AF_OUT_POWER)
    pipe.set_batch_uniform(1, 30000)
    got = pipe.run(x)
    xf = x.astype(np.float32) * np.float32(1.0 / 32768.0)
    assert frame_rel_err(got, po.stft_power(po.to_mono(xf, 2))) < TOL


@pytest.mark.parametrize("fmt", [0, 1])
def test_separate_channels(af, po, fmt):
    # rows [clip][channel][frame]: each channel is its own mono clip
    lens = [20000, 1024, 7000]
    clips = [synth_clip(20 + i, n, 2, 48000) for i, n in enumerate(lens)]
    if fmt:
        clips = [(c * 32767).astype(np.int16) for c in clips]
    pipe = af.FeaturePipeline(channels=2, sample_format=fmt, channel_mode=af._ffi.AF_CH_SEPARATE)
    total = pipe.set_batch(np.concatenate([[0], np.cumsum(lens)]))
    got = pipe.run(np.concatenate(clips))
    row = 0
    for c in clips:
        cf = c.astype(np.float32) * np.float32(1.0 / 32768.0) if fmt else c
        for chan in range(2):
            ref = po.features(np.ascontiguousarray(cf[chan::2]), 1, 48000.0)
            assert frame_rel_err(got[row:row + ref.shape[0]], ref) < TOL
            row += ref.shape[0]
    assert row == total == 2 * sum(pipe.n_frames(n) for n in lens)


def test_separate_channels_process_batch(af, po):
    # process_batch in separate mode: clip i's rows are [channel 0 frames | channel 1 frames]
    lens = [6000, 1024, 3000]
    frames = [af.AudioFrame.new(synth_clip(700 + i, n, 2, 48000), 48000, 2, 0) for i, n in enumerate(lens)]
    pipe = af.FeaturePipeline(channel_mode=af._ffi.AF_CH_SEPARATE)
    out = pipe.process_batch(frames)
    for f, m, n in zip(frames, out, lens):
        nf = po.stft_n_frames(n)
        assert m.data.shape[0] == 2 * nf
        x = f.samples.reshape(-1, 2)
        for c in range(2):
            ref = po.features(np.ascontiguousarray(x[:, c]), 1, 48000.0)
            assert frame_rel_err(m.data[c * nf:(c + 1) * nf], ref) < TOL


def test_separate_needs_stereo(af):
    with pytest.raises(af.AudioError):
        af.FeaturePipeline(channels=1, channel_mode=af._ffi.AF_CH_SEPARATE)


def test_log_mel(af, po):
    x = synth_clip(5, 30000, 2, 48000)
    x[:6000] = 0.0                                       # silent frames hit the 1e-10 floor
    pipe = af.FeaturePipeline(log_mel=True)
    pipe.set_batch_uniform(1, 30000)
    got = pipe.run(x)
    ref = np.maximum(po.features(x, 2, 48000.0), 1e-10)
    # judged where the tolerance is defined: on the mel values (per-frame norm); an f32 log10
    # value near -10 carries 1e-6 of rounding = 2.3e-6 relative on mel
    assert frame_rel_err(10.0 ** got.astype(np.float64), ref) < TOL
    assert np.max(np.abs(got[0] + 10.0)) < 2e-6          # silent frame: the 1e-10 floor


@pytest.mark.parametrize("scale,norm", [(1, 0), (1, 1), (0, 1)])
def test_slaney_filterbank(af, po, scale, norm):
    x = synth_clip(6, 20000, 1, 16000)
    pipe = af.FeaturePipeline(sample_rate=16000, channels=1, mel_scale=scale, mel_norm=norm)
    pipe.set_batch_uniform(1, 20000)
    W = po.mel_weights(16000.0, scale=scale, norm=norm)
    assert bits_equal(pipe.mel_weights(), W.astype(np.float32))
    ref = po.mel_apply(po.stft_power(x), W)
    assert frame_rel_err(pipe.run(x), ref) < TOL


@pytest.mark.parametrize("n_frames", [1, 15, 16, 17, 1000])
def test_mel_filterbank_apply(af, po, n_frames):
    rng = np.random.default_rng(n_frames)
    P = (rng.random((n_frames, 513)) ** 4).astype(np.float32) * 100.0
    fb = af.MelFilterbank(sample_rate=16000, n_mels=80)
    W = po.mel_weights(16000.0)
    assert bits_equal(fb.weights(), W.astype(np.float32))
    got = fb.apply(P)
    assert got.shape == (n_frames, 80)
    assert frame_rel_err(got, po.mel_apply(P, W)) < TOL


def test_stft_then_mel_equals_fused(af):
    # the two stand-alone stages reproduce the fused kernel's rows within f32 rounding
    x = synth_clip(8, 48000, 1, 16000)
    P = af.Stft().power(x)
    mel2 = af.MelFilterbank(sample_rate=16000).apply(P)
    pipe = af.FeaturePipeline(sample_rate=16000, channels=1)
    pipe.set_batch_uniform(1, 48000)
    assert frame_rel_err(mel2, pipe.run(x).astype(np.float64)) < 1e-6


@pytest.mark.parametrize("sr,ch,n_mels,hop,log", [(48000, 2, 20, 256, False), (48000, 1, 8, 160, False), (96000, 1, 13, 256, True),
                                                  (48000, 2, 2, 256, False), (44100, 2, 16, 256, False), (48000, 1, 1, 512, False)])
def test_wide_filterbanks_take_the_dense_contraction(af, po, sr, ch, n_mels, hop, log):
    """Few, wide filters (20 mels over 0 .. 24 kHz ...) do not fit the banded MFMA plan -- a group of filters must fit the
    K-steps of one wave --: the handle then runs the power kernel and a dense contraction with W instead of failing
    (round 2's AF_ERR_UNSUPPORTED).  Ragged batch, both kernels (hop 256 / general), log-mel, the stand-alone filterbank."""
    lens = [30000, 1024, 1023, 17000]
    clips = [synth_clip(40 + i, n, ch, sr) for i, n in enumerate(lens)]
    pipe = af.FeaturePipeline(sample_rate=sr, channels=ch, hop=hop, n_mels=n_mels, log_mel=log)
    total = pipe.set_batch(np.concatenate([[0], np.cumsum(lens)]))
    got = pipe.run(np.concatenate([x.reshape(-1) for x in clips]))
    assert got.shape == (total, n_mels)
    W = po.mel_weights(float(sr), n_mels=n_mels)
    assert bits_equal(pipe.mel_weights(), W.astype(np.float32))
    row = 0
    for x, n in zip(clips, lens):
        nf = pipe.n_frames(n)
        if nf:
            ref = po.features(x, ch, float(sr), hop=hop, n_mels=n_mels)
            g = got[row:row + nf].astype(np.float64)
            assert frame_rel_err(10.0 ** g if log else g, np.maximum(ref, 1e-10) if log else ref) < TOL
        row += nf
    assert row == total
    # the stand-alone filterbank on given power rows
    rng = np.random.default_rng(n_mels)
    P = (rng.random((37, 513)) ** 4).astype(np.float32) * 10.0
    fb = af.MelFilterbank(sample_rate=sr, n_mels=n_mels)
    assert frame_rel_err(fb.apply(P), po.mel_apply(P, W)) < TOL


@pytest.mark.parametrize("n_fft,hop,sr,ch,n_mels", [(512, 128, 16000, 1, 80), (512, 160, 16000, 2, 40), (256, 64, 8000, 1, 40),
                                                    (512, 256, 48000, 2, 80), (256, 256, 16000, 1, 23)])
def test_shorter_transforms(af, po, n_fft, hop, sr, ch, n_mels):
    """n_fft 512 / 256: the kernels transform the n_fft-sample frame zero-padded to 1024 points, whose every (1024 / n_fft)-th
    bin IS the n_fft-point DFT; frame counts, window, filterbank and rows are those of the n_fft-point definition
    (docs/OPS.md), against the oracle's n_fft-point transform.  Ragged batch incl. clips of n_fft and n_fft - 1 samples."""
    lens = [20000, n_fft, n_fft - 1, 7777]
    clips = [synth_clip(60 + i, n, ch, sr) for i, n in enumerate(lens)]
    offs = np.concatenate([[0], np.cumsum(lens)])
    flat = np.concatenate([x.reshape(-1) for x in clips])
    pipe = af.FeaturePipeline(sample_rate=sr, channels=ch, n_fft=n_fft, hop=hop, n_mels=n_mels)
    total = pipe.set_batch(offs)
    assert total == sum(po.stft_n_frames(n, n_fft, hop) for n in lens)
    got = pipe.run(flat)
    W = po.mel_weights(float(sr), n_fft=n_fft, n_mels=n_mels)
    assert bits_equal(pipe.mel_weights(), W.astype(np.float32))
    pw = af.FeaturePipeline(sample_rate=sr, channels=ch, n_fft=n_fft, hop=hop, n_mels=n_mels, output=af._ffi.AF_OUT_POWER)
    assert pw.set_batch(offs) == total and pw.row_len() == n_fft // 2 + 1
    power = pw.run(flat)
    row = 0
    for x, n in zip(clips, lens):
        nf = pipe.n_frames(n)
        if nf:
            mono = po.to_mono(x, ch)
            P = po.stft_power(mono, n_fft, hop)
            assert frame_rel_err(power[row:row + nf], P) < TOL
            assert frame_rel_err(got[row:row + nf], po.mel_apply(P, W)) < TOL
        row += nf
    assert row == total
    # the stand-alone stages on the same grid
    P = af.Stft(n_fft=n_fft, hop=hop, channels=ch).power(clips[0])
    assert P.shape[1] == n_fft // 2 + 1 and frame_rel_err(P, po.stft_power(po.to_mono(clips[0], ch), n_fft, hop)) < TOL
    fb = af.MelFilterbank(sample_rate=sr, n_fft=n_fft, n_mels=n_mels)
    assert frame_rel_err(fb.apply(P), po.mel_apply(P.astype(np.float64), W)) < TOL


def test_shorter_transform_streams_like_one_shot(af):
    """af_stream_* with n_fft = 512 (carry n_fft - hop = 384 frames): pushes + flush == one-shot, bit for bit; and a bank of
    20 mels at 48 kHz (the dense contraction inside the replayed graph)."""
    for kw, sr, ch in [(dict(n_fft=512, hop=128, n_mels=40, sample_rate=16000, channels=1), 16000, 1),
                       (dict(n_mels=20, sample_rate=48000, channels=2), 48000, 2)]:
        hop = kw.get("hop", 256)
        n = 16 * hop * 9 + 3333
        x = synth_clip(70, n, ch, sr)
        xf = x.reshape(n, ch)
        st = af.FeatureStream(step_frames=16 * hop * 2, **kw)
        rng = np.random.default_rng(6)
        rows, pos = [], 0
        while pos < n:
            k = int(rng.integers(1, 9000))
            rows.append(st.push(xf[pos:pos + k]))
            pos += k
        rows.append(st.flush())
        got = np.concatenate(rows)
        one = af.FeaturePipeline(**kw)
        one.set_batch_uniform(1, n)
        assert bits_equal(got, one.run(x.reshape(-1)))
