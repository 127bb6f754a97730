"""GPU parity of the fused STFT+mel path (through the C ABI) against the float64
oracle and the committed golden vectors.

Tolerance (docs/OPS.md, BASELINE.json north_star "within 1e-5 relative"):
per spectrogram frame  ||got - ref||_2 / ||ref||_2 <= 1e-5  against float64;
frame counts and output placement are exact.
"""
import os

import numpy as np
import pytest

from util import bits_equal, frame_rel_err, synth_clip

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-5


def run_clips(af, clips, ch, sr=48000, hop=256, n_mels=80, fmt=0):
    pipe = af.FeaturePipeline(sample_rate=sr, channels=ch, hop=hop, n_mels=n_mels, sample_format=fmt)
    offs = np.zeros(len(clips) + 1, np.int64)
    for i, c in enumerate(clips):
        offs[i + 1] = offs[i] + c.size // ch
    total = pipe.set_batch(offs)
    dt = np.int16 if fmt else np.float32
    pcm = np.concatenate(clips) if clips else np.zeros(0, dt)
    mel = pipe.run(pcm)
    assert mel.shape == (total, n_mels)
    return pipe, mel


def test_golden_48k_stereo(af):
    g = np.load(os.path.join(GOLD, "features_48k_stereo.npz"))
    _, mel = run_clips(af, [g["pcm"]], 2)
    assert mel.shape == g["mel"].shape
    assert frame_rel_err(mel, g["mel"]) < TOL


def test_golden_16k_mono_config1(af):
    # BASELINE config 1: 10 s mono 16 kHz -> 622 frames (count exact), values on the fixture part
    g = np.load(os.path.join(GOLD, "features_16k_mono.npz"))
    pipe, mel = run_clips(af, [g["pcm"]], 1, sr=16000)
    assert pipe.n_frames(int(g["n_samples_full"])) == 622
    assert frame_rel_err(mel, g["mel"]) < TOL


@pytest.mark.parametrize("ch", [1, 2])
@pytest.mark.parametrize("hop", [256, 128, 512, 320, 64, 1024])
def test_ragged_batch_vs_oracle(af, po, ch, hop):
    # empty, shorter-than-a-frame, exactly one frame, unit boundaries (16 frames), long
    lens = [0, 1023, 1024, 1024 + hop * 15, 1024 + hop * 16, 1024 + hop * 16 - 1, 7777, 50000, 3]
    clips = [synth_clip(i, n, ch, 48000) for i, n in enumerate(lens)]
    pipe, mel = run_clips(af, clips, ch, hop=hop)
    row = 0
    for n, x in zip(lens, clips):
        ref = po.features(x, ch, 48000.0, hop=hop)
        nf = po.stft_n_frames(n, 1024, hop)
        assert ref.shape[0] == nf == pipe.n_frames(n)            # bookkeeping is exact
        assert frame_rel_err(mel[row:row + nf], ref) < TOL
        row += nf
    assert row == mel.shape[0]
    assert np.all(np.isfinite(mel))


@pytest.mark.parametrize("ch", [1, 2])
@pytest.mark.parametrize("fmt", [0, 1])
def test_hop256_kernel_aligned_ragged(af, po, ch, fmt):
    """Clip bases on 16-byte boundaries at hop 256 select the 8-wave kernel (features_v2.hip):
    empty / short / unit-boundary / long clips in all four PCM formats against the oracle and
    against the general kernel (af_debug_set "features.kernel" = 1) on the same input."""
    lens = [0, 1016, 1024, 1024 + 256 * 15, 1024 + 256 * 16, 1024 + 256 * 16 - 8, 7776, 50000, 8,
            1024 + 256 * 47, 123456]
    clips = [synth_clip(i, n, ch, 48000) for i, n in enumerate(lens)]
    if fmt:
        clips = [np.round(c * 32767.0).astype(np.int16) for c in clips]
    lib = af.library()
    pipe, mel = run_clips(af, clips, ch, fmt=fmt)
    try:
        assert lib.af_debug_set(b"features.kernel", 1) == 0
        _, mel_g = run_clips(af, clips, ch, fmt=fmt)
    finally:
        lib.af_debug_set(b"features.kernel", 0)
    row = 0
    for n, x in zip(lens, clips):
        xf = x.astype(np.float32) / np.float32(32768.0) if fmt else x
        ref = po.features(xf, ch, 48000.0)
        nf = ref.shape[0]
        assert nf == pipe.n_frames(n)
        assert frame_rel_err(mel[row:row + nf], ref) < TOL
        assert frame_rel_err(mel[row:row + nf], mel_g[row:row + nf]) < TOL
        row += nf
    assert row == mel.shape[0] and np.all(np.isfinite(mel))


@pytest.mark.parametrize("ch", [3, 4, 6])
def test_more_than_two_channels(af, po, ch):
    # AudioFrame::to_mono (capture.rs:30-42) takes any channel count: to_mono runs first, then the mono kernel
    lens = [20000, 1024, 4864 + 256 * 20, 0, 7777]
    clips = [synth_clip(800 + i, n, ch, 48000) for i, n in enumerate(lens)]
    pipe, mel = run_clips(af, clips, ch)
    row = 0
    for n, x in zip(lens, clips):
        ref = po.features(x, ch, 48000.0)
        nf = ref.shape[0]
        assert nf == pipe.n_frames(n)
        assert frame_rel_err(mel[row:row + nf], ref) < TOL
        row += nf
    assert row == mel.shape[0]
    with pytest.raises(af.AudioError):
        af.FeaturePipeline(channels=4, sample_format=af._ffi.AF_PCM_I16)


def test_unaligned_clip_offsets_take_the_scalar_path(af, po):
    # odd frame offsets break 16-byte alignment of the clip bases: same results required
    lens = [3001, 2047, 4099]
    clips = [synth_clip(20 + i, n, 2, 48000) for i, n in enumerate(lens)]
    _, mel = run_clips(af, clips, 2)
    row = 0
    for x in clips:
        ref = po.features(x, 2, 48000.0)
        assert frame_rel_err(mel[row:row + ref.shape[0]], ref) < TOL
        row += ref.shape[0]


@pytest.mark.parametrize("ch", [1, 2])
def test_i16_input(af, po, ch):
    x = synth_clip(4, 20000, ch, 48000)
    xi = np.round(x * 32767.0).astype(np.int16)
    _, mel = run_clips(af, [xi], ch, fmt=af._ffi.AF_PCM_I16)
    ref = po.features(xi.astype(np.float32) / np.float32(32768.0), ch, 48000.0)   # docs/OPS.md
    assert frame_rel_err(mel, ref) < TOL


@pytest.mark.parametrize("sr,n_mels", [(16000, 80), (48000, 40), (44100, 128), (48000, 30), (22050, 27),
                                        (96000, 64)])
def test_other_filterbanks(af, po, sr, n_mels):
    x = synth_clip(6, 30000, 1, sr)
    pipe, mel = run_clips(af, [x], 1, sr=sr, n_mels=n_mels)
    ref = po.features(x, 1, float(sr), n_mels=n_mels)
    assert frame_rel_err(mel, ref) < TOL
    # the device's filterbank is the oracle's, rounded to f32
    assert bits_equal(pipe.mel_weights(), po.mel_weights(float(sr), n_mels=n_mels).astype(np.float32))


def test_process_batch_surface(af, po):
    frames = [af.AudioFrame.new(synth_clip(30 + i, n, 2, 48000), 48000, 2, 0)
              for i, n in enumerate([5000, 1024, 100])]
    pipe = af.FeaturePipeline()
    out = pipe.process_batch(frames)
    assert [m.data.shape[0] for m in out] == [16, 1, 0]
    for f, m in zip(frames, out):
        assert frame_rel_err(m.data, po.features(f.samples, 2, 48000.0)) < TOL
    with pytest.raises(af.AudioError):
        pipe.process_batch([af.AudioFrame.new(np.zeros(4000, np.float32), 16000, 1, 0)])


def test_too_wide_filters_take_the_dense_path(af, po):
    # rounds 1-2 reported AF_ERR_UNSUPPORTED here (4 consecutive filters may span at most 256 bins in the banded MFMA
    # plan); round 3 falls back to power rows + a dense contraction (tests/test_gpu_stft_mel_modes.py covers it)
    x = synth_clip(31, 20000, 1, 48000)
    pipe = af.FeaturePipeline(sample_rate=48000, channels=1, n_mels=8)
    pipe.set_batch_uniform(1, 20000)
    assert frame_rel_err(pipe.run(x), po.features(x, 1, 48000.0, n_mels=8)) < TOL


def test_invalid_configs_are_rejected(af):
    for kw in [dict(n_fft=2048), dict(n_fft=16), dict(n_fft=1025), dict(channels=17), dict(channels=0), dict(hop=30), dict(hop=258), dict(n_mels=0),
               dict(n_mels=200), dict(sample_rate=0), dict(f_min=30000.0)]:
        with pytest.raises(af.AudioError):
            af.FeaturePipeline(**kw)


def test_silence_and_extremes(af, po):
    z = np.zeros(8000, np.float32)
    _, mel = run_clips(af, [z], 2)
    assert np.all(mel == 0.0)
    full = np.ones(8000, np.float32)            # clipped DC at full scale
    _, mel = run_clips(af, [full], 2)
    ref = po.features(full, 2, 48000.0)
    # a constant signal under a Hann window leaves energy only in bins 0,1: compare absolutely
    assert np.max(np.abs(mel - ref)) <= 1e-5 * np.max(ref)


# ---- full-size properties (BASELINE config 2 shapes) on device-resident data -------------
def test_full_size_properties(af, po):
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    n_clips, n, ch = 256, 1440000, 2              # BASELINE config 2 in full: 256 x 30 s stereo 48 kHz (2.95 GB)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    pcm = torch.rand(n_clips, n, ch, device=dev, generator=gen).mul_(0.5).sub_(0.25)
    t = torch.arange(n, device=dev, dtype=torch.float32) / 48000.0
    pcm[:, :, 0] += 0.3 * torch.sin(2 * torch.pi * 440.0 * t)
    pcm[:, :, 1] += 0.2 * torch.sin(2 * torch.pi * 3000.0 * t)
    pcm[7] = pcm[3]                               # duplicate clips must give identical rows
    pcm[255] = pcm[0]
    pipe = af.FeaturePipeline()
    total = pipe.set_batch_uniform(n_clips, n)
    nf = pipe.n_frames(n)
    assert nf == 5622 and total == n_clips * nf    # exact bookkeeping
    mel = torch.full((total, 80), float("nan"), device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    pipe.run_device(pcm.data_ptr(), mel.data_ptr(), stream)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(mel).all())         # every row written
    m = mel.view(n_clips, nf, 80)
    # 1. a clip's result does not depend on where it sits in the batch (bit-exact)
    assert bits_equal(m[7], m[3]) and bits_equal(m[255], m[0])
    # 2. determinism: a second launch reproduces every bit
    mel2 = torch.empty_like(mel)
    pipe.run_device(pcm.data_ptr(), mel2.data_ptr(), stream)
    torch.cuda.synchronize()
    assert bits_equal(mel, mel2)
    # 3. power-of-two scaling is exact: x*2 -> mel*4
    pcm2 = pcm[:4] * 2.0
    pipe4 = af.FeaturePipeline()
    pipe4.set_batch_uniform(4, n)
    mel4 = torch.empty(4 * nf, 80, device=dev)
    pipe4.run_device(pcm2.data_ptr(), mel4.data_ptr(), stream)
    torch.cuda.synchronize()
    assert bits_equal(mel4.view(4, nf, 80), m[:4] * 4.0)
    # 4. sampled clips against the float64 oracle
    for c in (0, 131, 254):
        ref = po.features(pcm[c].cpu().numpy().reshape(-1), ch, 48000.0)
        assert frame_rel_err(m[c].cpu().numpy(), ref) < TOL


def test_host_buffer_path_is_chunked_and_identical(af):
    """af_features_run cuts a large contiguous batch into clip-range chunks and overlaps H2D / kernel /
    D2H on three streams: same rows as the device-resident launch, from pageable and page-locked memory,
    with ragged clips and empty ones in the batch."""
    torch = pytest.importorskip("torch")
    import ctypes as C
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(9)
    lens = [int(x) * 8 for x in rng.integers(30000, 60000, size=30)] + [0, 512, 1024]     # ~86 MB of stereo f32
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    n = int(offs[-1])
    pinned = torch.empty(n * 2, dtype=torch.float32).pin_memory()
    pinned.uniform_(-0.4, 0.4, generator=torch.Generator().manual_seed(3))
    pipe = af.FeaturePipeline()
    total = pipe.set_batch(offs)
    ref = torch.empty(total, 80, device=dev)
    d = pinned.to(dev)
    pipe.run_device(d.data_ptr(), ref.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ref = ref.cpu().numpy()
    got_pageable = pipe.run(pinned.numpy().copy())                   # a fresh pageable copy
    assert bits_equal(got_pageable, ref)
    out = torch.empty(total, 80).pin_memory()
    st = af.library().af_features_run(pipe._h, C.c_void_p(pinned.data_ptr()), pinned.numel(),
                                      C.c_void_p(out.data_ptr()), out.numel())
    assert st == 0 and bits_equal(out.numpy(), ref)
    # separate channels: a clip's two table entries stay in one chunk
    sep = af.FeaturePipeline(channel_mode=af._ffi.AF_CH_SEPARATE)
    tot2 = sep.set_batch(offs)
    ref2 = torch.empty(tot2, 80, device=dev)
    sep.run_device(d.data_ptr(), ref2.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert bits_equal(sep.run(pinned.numpy()), ref2.cpu().numpy())


def test_many_ragged_clips_stress(af, po):
    # 3000 clips of random length (a third of them too short for a single frame), scalar-path
    # offsets: placement and values of every clip's rows; a sample of clips against the oracle
    rng = np.random.default_rng(2024)
    lens = rng.integers(0, 5000, size=3000)
    lens[rng.integers(0, 3000, size=50)] = 1024                     # exactly one frame
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    pcm = (rng.standard_normal(int(offs[-1])) * 0.2).astype(np.float32)
    pipe = af.FeaturePipeline(sample_rate=16000, channels=1)
    total = pipe.set_batch(offs)
    frames = np.array([po.stft_n_frames(int(n)) for n in lens])
    assert total == int(frames.sum())
    mel = pipe.run(pcm)
    assert np.isfinite(mel).all()
    rows = np.concatenate([[0], np.cumsum(frames)])
    for c in rng.choice(3000, size=40, replace=False):
        if frames[c] == 0:
            continue
        ref = po.features(pcm[offs[c]:offs[c + 1]], 1, 16000.0)
        assert frame_rel_err(mel[rows[c]:rows[c + 1]], ref) < TOL


def test_one_shot_pipeline_run(af, po):
    # af_pipeline_run: handle-free form of create / set_batch / run / destroy
    import ctypes as C
    lens = [3000, 1024, 500]
    clips = [synth_clip(600 + i, n, 2, 48000) for i, n in enumerate(lens)]
    pcm = np.concatenate(clips)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    d = af._ffi.BatchDesc()
    af.library().af_features_default_config(C.byref(d.cfg))
    d.n_clips = len(lens)
    d.clip_offsets = offs.ctypes.data
    total = C.c_size_t()
    assert af.library().af_pipeline_run(C.byref(d), None, 0, None, 0, C.byref(total)) == 0
    assert total.value == sum(po.stft_n_frames(n) for n in lens)
    out = np.empty((total.value, 80), np.float32)
    assert af.library().af_pipeline_run(C.byref(d), pcm.ctypes.data, pcm.size, out.ctypes.data, out.size, C.byref(total)) == 0
    ref = np.concatenate([po.features(c, 2, 48000.0) for c in clips])
    assert frame_rel_err(out, ref) < TOL
    # a PCM buffer shorter than the batch describes is refused, not read out of bounds
    st = af.library().af_pipeline_run(C.byref(d), pcm.ctypes.data, pcm.size - 1, out.ctypes.data, out.size,
                                      C.byref(total))
    assert st == af._ffi.AF_ERR_BUFFER_TOO_SMALL
    pipe = af.FeaturePipeline()
    pipe.set_batch(offs)
    with pytest.raises(af.AudioError) as e:
        pipe.run(pcm[:-2])
    assert e.value.status == af._ffi.AF_ERR_BUFFER_TOO_SMALL


@pytest.mark.parametrize("ch,fmt", [(2, 0), (1, 0), (2, 1), (1, 1)])
def test_hop256_kernel_on_clips_off_16_byte_boundaries(af, po, ch, fmt):
    """Clip bases that are only dword-aligned (odd frame offsets) stay on the hop-256 kernel (its 16-byte global loads
    need no alignment beyond the sample's own: since round 5 that includes i16 mono clips on odd frames).  Rows equal
    the general kernel's within rounding and the oracle's within tolerance; a clip's rows do not depend on where it
    starts."""
    import ctypes as C
    lens = [30001, 5003, 12345, 40007]
    clips = [synth_clip(500 + i, n, ch, 48000) for i, n in enumerate(lens)]
    if fmt:
        clips = [np.round(c * 32767.0).astype(np.int16) for c in clips]
    pcm = np.concatenate(clips)
    off = np.concatenate([[0], np.cumsum(lens)])
    pipe = af.FeaturePipeline(channels=ch, sample_format=fmt)
    pipe.set_batch(off)
    got = pipe.run(pcm)
    name = af.library().af_features_kernel_name().decode()
    lib = af.library()
    assert lib.af_features_set_option(pipe._h, b"kernel", 1) == 0
    gen = pipe.run(pcm)
    assert frame_rel_err(got, gen) < 2e-6
    row = 0
    for x, n in zip(clips, lens):
        nf = pipe.n_frames(n)
        xf = x.astype(np.float32) / np.float32(32768.0) if fmt else x
        assert frame_rel_err(got[row:row + nf], po.features(xf, ch, 48000.0)) < 1e-5
        # the same clip alone, at offset 0 (16-byte aligned): identical rows on the hop-256 kernel
        alone = af.FeaturePipeline(channels=ch, sample_format=fmt)
        alone.set_batch_uniform(1, n)
        assert bits_equal(alone.run(x), got[row:row + nf])
        row += nf


@pytest.mark.parametrize("kw", [dict(), dict(n_mels=64, sample_rate=16000), dict(mel_scale=1, mel_norm=1), dict(n_fft=512),
                                dict(sample_format=1), dict(channel_mode=1), dict(log_mel=True),
                                dict(n_mels=40), dict(n_mels=128, sample_rate=44100), dict(n_mels=23, sample_rate=8000)],
                         ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()) or "default")
def test_hop256_mel_plans_agree(af, po, kw):
    """The hop-256 kernel has two mel phases: 16 blocks of 4 x 4 x 1 (the default where the plan fits: n_mels a multiple
    of 4 and <= 16 pieces per half of the bands) and 16 x 16 x 4 tiles (`af_features_set_option(h, "mel4", 0)`, and every
    other filterbank).  Same rows within float32 summation order, both inside the oracle tolerance; ragged clips (one
    ends 8 samples short of a unit, one is a single frame, one is shorter than n_fft)."""
    lib = af.library()
    sr, ch = kw.get("sample_rate", 48000), 2
    lens = [20000, 1024 + 256 * 16 - 8, 1024, 900, 50000]
    clips = [synth_clip(300 + i, n, ch, sr) for i, n in enumerate(lens)]
    if kw.get("sample_format"):
        clips = [np.round(c * 32767.0).astype(np.int16) for c in clips]
    x = np.concatenate([c.reshape(-1) for c in clips])
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    out = []
    for m4 in (-1, 0):
        p = af.FeaturePipeline(channels=ch, **kw)
        assert lib.af_features_set_option(p._h, b"mel4", m4) == 0
        p.set_batch(off)
        out.append(p.run(x))
    assert np.isfinite(out[0]).all() and out[0].shape == out[1].shape
    if kw.get("log_mel"):
        assert float(np.abs(out[0] - out[1]).max()) < 2e-5          # log10 of sums that differ in the last bits
        return
    assert frame_rel_err(out[0], np.asarray(out[1], np.float64)) < 2e-6
    if kw.get("channel_mode"):
        return
    W = po.mel_weights(float(sr), kw.get("n_fft", 1024), kw.get("n_mels", 80), scale=kw.get("mel_scale", 0), norm=kw.get("mel_norm", 0))
    row = 0
    for c, n in zip(clips, lens):
        cf = c.astype(np.float32) / np.float32(32768.0) if kw.get("sample_format") else c
        ref = po.mel_apply(po.stft_power(po.to_mono(cf, ch), kw.get("n_fft", 1024), 256), W)
        if ref.shape[0]:
            for o in out:
                assert frame_rel_err(o[row:row + ref.shape[0]], ref) < TOL
        row += ref.shape[0]
    assert row == out[0].shape[0]


def test_kernel_describe_names_the_family_and_the_round5_plans(af):
    """`af_features_kernel_describe`: which kernel family serves a handle and the shape of the 4 x 4 x 1 mel plan / the
    matrix-product DFT's tables -- the documented cases (DESIGN.md sections 6 and 6g)."""
    import ctypes as C
    lib = af.library()

    def describe(**kw):
        p = af.FeaturePipeline(**kw)
        info = (C.c_int32 * 12)()
        assert lib.af_features_kernel_describe(p._h, info) == 1
        return list(info)

    d = describe()                                               # BASELINE config 2's handle
    assert d[0] == 2 and d[1] == 1 and d[2] == 8 and d[3] == 8 and 16 >= d[4] >= 10 and 16 >= d[5] >= 10
    d = describe(sample_rate=16000, n_mels=64)
    assert d[0] == 2 and d[1] == 1 and d[2] <= 8 and d[3] <= 8
    assert describe(n_mels=40)[:2] == [2, 0]                     # too few, too wide bands for 16 blocks at 32 steps: 16 x 16 x 4 tiles
    assert describe(sample_rate=44100, n_mels=128)[:2] == [2, 0]
    assert describe(n_mels=23, sample_rate=8000)[1] == 0         # not a multiple of 4
    assert describe(hop=160)[0] == 1                             # the general fused kernel
    d = describe(sample_rate=16000, channels=1, n_fft=400, hop=160)
    assert d[0] == 4 and d[1] == 0 and d[6:10] == [13, 13, 5, 2] and d[10] == 16 * (2 * 16 * 13 + 4) * 4
    d = describe(sample_rate=16000, channels=1, n_fft=100, hop=64)
    assert d[0] == 4 and d[6] == 4                               # at least four K groups (the kernel walks chunks of four)


@pytest.mark.parametrize("kw", [dict(), dict(n_mels=40), dict(sample_rate=16000, channels=1, n_fft=400, hop=160), dict(hop=160)],
                         ids=["hop256-mel4", "hop256-tiles", "dft-400", "general"])
def test_output_rows_at_a_pointer_that_is_only_4_byte_aligned(af, kw):
    """`af_features_run_device` with a row buffer that starts 4 bytes off a 16-byte boundary: the kernels fall back from their
    16-byte stores to single floats; same rows, nothing written in front of or behind them."""
    import torch
    dev = torch.device("cuda:0")
    ch, sr = kw.get("channels", 2), kw.get("sample_rate", 48000)
    n = 40000
    x = torch.from_numpy(synth_clip(91, n, ch, sr)).to(dev)
    pipe = af.FeaturePipeline(**kw)
    total = pipe.set_batch_uniform(1, n)
    nm = kw.get("n_mels", 80)
    s = torch.cuda.current_stream().cuda_stream
    ref = torch.empty(total, nm, device=dev)
    pipe.run_device(x.data_ptr(), ref.data_ptr(), s)
    buf = torch.full((total * nm + 8,), -7.0, device=dev)
    pipe.run_device(x.data_ptr(), buf.data_ptr() + 4, s)
    torch.cuda.synchronize()
    got = buf[1:1 + total * nm].reshape(total, nm)
    assert bits_equal(got.cpu().numpy(), ref.cpu().numpy())
    assert float(buf[0]) == -7.0 and bool((buf[1 + total * nm:] == -7.0).all())


def test_work_pool_gives_the_same_rows(af):
    """The hop-256 kernel deals the last eighth of every workgroup's unit range through a device-wide counter (batches of at
    least 64 units per workgroup; `af_features_set_option(h, "pool", 0)` keeps the static ranges).  Which workgroup computes a
    unit, and whether it loads its signal tile whole or shifts the previous one, does not change a bit of its rows; two
    launches of the pooled handle agree with each other as well (the deal differs from launch to launch)."""
    import torch
    dev = torch.device("cuda:0")
    lib = af.library()
    clips, n = 100, 1440000
    g = torch.Generator(device="cpu"); g.manual_seed(5)
    x = ((torch.rand(4, n, 2, generator=g) * 2 - 1) * 0.3).to(dev).repeat(25, 1, 1).contiguous()
    s = torch.cuda.current_stream().cuda_stream
    out = []
    for pool in (0, -1, -1):
        p = af.FeaturePipeline()
        assert lib.af_features_set_option(p._h, b"pool", pool) == 0
        total = p.set_batch_uniform(clips, n)
        mel = torch.empty(total, 80, device=dev)
        p.run_device(x.data_ptr(), mel.data_ptr(), s)
        torch.cuda.synchronize()
        out.append(mel)
    assert torch.equal(out[0].view(torch.int32), out[1].view(torch.int32))
    assert torch.equal(out[1].view(torch.int32), out[2].view(torch.int32))
    nf = total // clips
    m = out[1].reshape(clips, nf, 80)
    assert torch.equal(m[:4].view(torch.int32), m[96:100].view(torch.int32))     # the batch repeats four clips


def test_mel_plan_arrangement_does_not_change_a_bit(af, monkeypatch):
    """`build_plan4` rearranges the 4 x 4 x 1 plan's pieces (rows, order in a row, first bin of each piece's window) against LDS
    bank conflicts; `AF_M4_NO_ARRANGE=1` at handle creation keeps the first-fit layout.  A piece's products are summed in bin order
    either way and the pieces of a band in the same order, so the rows are the same bits; the conflict score drops."""
    import ctypes as C
    lib = af.library()
    x = synth_clip(17, 60000, 2, 48000)

    def run():
        p = af.FeaturePipeline()
        info = (C.c_int32 * 12)()
        assert lib.af_features_kernel_describe(p._h, info) == 1
        p.set_batch_uniform(1, 60000)
        return p.run(x), info[11]

    arranged, score = run()
    monkeypatch.setenv("AF_M4_NO_ARRANGE", "1")
    first_fit, score_ff = run()
    monkeypatch.delenv("AF_M4_NO_ARRANGE")
    assert bits_equal(arranged, first_fit)
    assert score <= 8 and score == score_ff              # (the score is the search's result either way; the layout differs)
