"""The matrix-core DFT kernel (features_v3.hip; `af_features_set_option(h, "kernel", 3)`): the same contract as the
other fused kernels -- to_mono (capture.rs:30-42), Hann STFT 1024 / 256, |X|^2, mel (docs/OPS.md, build-defined) --
with the transform as two f16-pair matrix products.  Parity against the float64 oracle and the register-FFT kernel,
placement invariance, the per-unit scale over large level steps, the limits of what it serves."""
import numpy as np
import pytest

from util import bits_equal, frame_rel_err, synth_clip

pytestmark = pytest.mark.gpu
TOL = 1e-5                      # north_star: within 1e-5 relative of the CPU reference, per frame


def _run(af, clips, ch, fmt=0, kernel=3, **kw):
    pipe = af.FeaturePipeline(sample_rate=kw.pop("sample_rate", 48000), channels=ch, hop=256, sample_format=fmt, **kw)
    assert af.library().af_features_set_option(pipe._h, b"kernel", kernel) == 0
    offs = np.zeros(len(clips) + 1, np.int64)
    for i, c in enumerate(clips):
        offs[i + 1] = offs[i] + c.size // ch
    pipe.set_batch(offs)
    return pipe, pipe.run(np.concatenate(clips))


@pytest.mark.parametrize("ch", [1, 2])
@pytest.mark.parametrize("fmt", [0, 1])
def test_v3_ragged_all_formats(af, po, ch, fmt):
    """Empty / short / unit-boundary / long clips in all four PCM formats: oracle parity, agreement with the
    register-FFT kernel, exact frame counts; a clip alone (other batch position) gives the same bits."""
    lens = [0, 1016, 1024, 1024 + 256 * 15, 1024 + 256 * 16, 1024 + 256 * 16 - 8, 7776, 50000, 8, 1024 + 256 * 47, 123457]
    clips = [synth_clip(i, n, ch, 48000) for i, n in enumerate(lens)]
    if fmt:
        clips = [np.round(c * 32767.0).astype(np.int16) for c in clips]
    pipe, mel = _run(af, clips, ch, fmt, 3)
    _, mel2 = _run(af, clips, ch, fmt, 2)
    row = 0
    for n, x in zip(lens, clips):
        xf = x.astype(np.float32) / np.float32(32768.0) if fmt else x
        ref = po.features(xf, ch, 48000.0)
        nf = ref.shape[0]
        assert nf == pipe.n_frames(n)
        if nf:
            assert frame_rel_err(mel[row:row + nf], ref) < TOL
            assert frame_rel_err(mel[row:row + nf], np.asarray(mel2[row:row + nf], np.float64)) < 2e-6
            if n in (50000, 123457) and not (ch == 1 and fmt == 1 and (sum(lens[:lens.index(n)]) & 1)):
                _, alone = _run(af, [x], ch, fmt, 3)
                assert bits_equal(alone, mel[row:row + nf])      # the unit scale depends on the unit's samples only
        row += nf
    assert row == mel.shape[0] and np.all(np.isfinite(mel))


def test_v3_level_steps_inside_a_unit(af, po):
    """The tile of a 16-frame unit is scaled by a power of two of its largest sample before it becomes f16 pairs: frames
    far below the unit's peak (here next to full-scale noise inside one unit) still meet the tolerance."""
    rng = np.random.default_rng(5)
    n = 1024 + 256 * 63
    half = n // 2
    for name, x in [("x 1e-6", 1e-6 * rng.standard_normal(n)), ("x 1e+4", 1e4 * rng.standard_normal(n)),
                    ("80 dB", np.concatenate([rng.standard_normal(half), 1e-4 * rng.standard_normal(n - half)])),
                    ("120 dB", np.concatenate([rng.standard_normal(half), 1e-6 * rng.standard_normal(n - half)])),
                    ("silence + tone", np.concatenate([np.zeros(half), np.sin(0.05 * np.arange(n - half))])),
                    ("dc", np.full(n, 0.7))]:
        x = x.astype(np.float32)
        ref = po.features(x, 1, 48000.0)
        _, got = _run(af, [x], 1, 0, 3)
        g = np.asarray(got, np.float64)
        nz = np.linalg.norm(ref, axis=1) > 0
        err = np.linalg.norm(g[nz] - ref[nz], axis=1) / np.linalg.norm(ref[nz], axis=1)
        assert err.max() < TOL, (name, err.max())
        assert np.all(g[~nz] == 0.0), name
    _, z = _run(af, [np.zeros(n, np.float32)], 1, 0, 3)
    assert np.all(z == 0.0)


def test_v3_non_finite_samples_stay_local(af, po):
    """A NaN and an Inf sample: the frames whose support contains one come out non-finite, every other frame -- also those
    in the same 16-frame unit, whose tile scale must not be driven by the Inf -- matches the oracle of the clean signal."""
    n = 1024 + 256 * 80
    x = synth_clip(77, n, 1, 48000).copy()
    clean = po.features(x, 1, 48000.0)
    for pos, bad in ((5000, np.float32(np.nan)), (12345, np.float32(np.inf)), (17000, np.float32(-np.inf))):
        y = x.copy()
        y[pos] = bad
        _, got = _run(af, [y], 1, 0, 3)
        got = np.asarray(got, np.float64)
        touched = np.array([(256 * f <= pos < 256 * f + 1024) for f in range(got.shape[0])])
        assert not np.any(np.isfinite(got[touched]).all(axis=1))
        ok = ~touched
        assert np.all(np.isfinite(got[ok]))
        err = np.linalg.norm(got[ok] - clean[ok], axis=1) / np.linalg.norm(clean[ok], axis=1)
        assert err.max() < TOL, (pos, err.max())


# filterbanks / modes the kernel's plan is documented to cover (separate channels, 40-128 filters at
# 16-48 kHz, HTK and Slaney, n_fft 1024 / 512 / 256): it must serve them, and within the tolerance of the ORACLE
V3_COVERED = [dict(channel_mode=1), dict(n_mels=40), dict(n_mels=128, sample_rate=44100), dict(n_fft=512), dict(n_fft=256),
              dict(mel_scale=1, mel_norm=1), dict(n_mels=64, sample_rate=16000, channels=1)]


@pytest.mark.parametrize("kw", V3_COVERED)
def test_v3_other_configurations(af, po, kw):
    """Separate channels, other filterbanks, shorter frames: the rows meet the float64 oracle's within 1e-5 per frame (the
    contract), and the register-FFT kernel's within 2e-6.  A configuration on this list that the kernel refused would be a
    regression of its mel plan, so nothing is skipped."""
    kw = dict(kw)
    ch = kw.pop("channels", 2)
    sr = kw.get("sample_rate", 48000)
    x = [synth_clip(40 + i, 20000 + 4096 * i, ch, sr) for i in range(3)]
    _, a = _run(af, x, ch, 0, 3, **dict(kw))
    _, b = _run(af, x, ch, 0, 2, **dict(kw))
    assert a.shape == b.shape and frame_rel_err(a, np.asarray(b, np.float64)) < 2e-6
    # the oracle, clip by clip (separate channels: a clip's rows are channel 0's spectrogram, then channel 1's)
    n_fft = kw.get("n_fft", 1024)
    W = po.mel_weights(float(sr), n_fft, kw.get("n_mels", 80), scale=kw.get("mel_scale", 0), norm=kw.get("mel_norm", 0))

    def oracle(mono):
        return po.mel_apply(po.stft_power(mono, n_fft, 256), W)

    row = 0
    for c in x:
        if kw.get("channel_mode"):
            refs = [oracle(np.ascontiguousarray(c.reshape(-1, 2)[:, k])) for k in (0, 1)]
        else:
            refs = [oracle(po.to_mono(c, ch))]
        for ref in refs:
            nf = ref.shape[0]
            assert frame_rel_err(a[row:row + nf], ref) < TOL
            row += nf
    assert row == a.shape[0]


def test_v3_filterbank_outside_its_plan_is_refused_by_name(af):
    """few, wide filters take the dense contraction (power rows + a second launch): the matrix-core kernel says so"""
    pipe = af.FeaturePipeline(sample_rate=48000, channels=2, n_mels=16)
    pipe.set_batch_uniform(1, 30000)
    assert af.library().af_features_set_option(pipe._h, b"kernel", 3) == 0
    with pytest.raises(af.AudioError) as e:
        pipe.run(synth_clip(3, 30000, 2, 48000))
    assert "kernel = 3" in str(e.value) and ("dense" in str(e.value) or "plan" in str(e.value))


def test_v3_refuses_what_it_does_not_serve(af):
    """Power output, log-mel and other hops stay with the other kernels: kernel = 3 is an error there, never a silent
    fallback; the default choice is unaffected."""
    x = synth_clip(3, 30000, 2, 48000)
    for kw in (dict(output=1), dict(log_mel=1), dict(hop=160)):
        pipe = af.FeaturePipeline(sample_rate=48000, channels=2, **({"hop": 256} | kw))
        pipe.set_batch_uniform(1, 30000)
        ok = pipe.run(x)
        assert np.all(np.isfinite(ok))
        assert af.library().af_features_set_option(pipe._h, b"kernel", 3) == 0
        with pytest.raises(af.AudioError):
            pipe.run(x)
        assert af.library().af_features_set_option(pipe._h, b"kernel", -1) == 0
        assert bits_equal(pipe.run(x), ok)


def test_v3_config2_shard_against_oracle(af, po):
    """A slice of BASELINE config 2 (8 x 30 s stereo 48 kHz, device-resident) through the matrix-core kernel: three
    clips against the oracle, all rows finite."""
    torch = pytest.importorskip("torch")
    n_clips, n = 8, 1440000
    gen = torch.Generator().manual_seed(4)
    pcm = ((torch.rand(n_clips, n, 2, generator=gen) * 2 - 1) * 0.3).cuda()
    pipe = af.FeaturePipeline(sample_rate=48000, channels=2)
    assert af.library().af_features_set_option(pipe._h, b"kernel", 3) == 0
    total = pipe.set_batch_uniform(n_clips, n)
    mel = torch.empty(total, 80, device="cuda")
    pipe.run_device(pcm.data_ptr(), mel.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    nf = pipe.n_frames(n)
    assert bool(torch.isfinite(mel).all())
    for c in (0, 3, 7):
        ref = po.features(pcm[c].cpu().numpy().reshape(-1), 2, 48000.0)
        assert frame_rel_err(mel[c * nf:(c + 1) * nf].cpu().numpy(), ref) < TOL
