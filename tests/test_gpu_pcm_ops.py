"""GPU parity of the reference's small PCM ops through the C ABI:
to_mono (capture.rs:30-42), PCM16 quantiser + base64 (websocket.rs:246-254),
VAD energy + state machine (vad.rs:97-176).  Integer/byte results are bit-exact;
to_mono is bit-exact (same f32 operations in the same order)."""
import base64

import numpy as np
import pytest

from util import bits_equal, synth_clip

pytestmark = pytest.mark.gpu


# ---- to_mono -------------------------------------------------------------------------
def test_to_mono_reference_vectors(af):
    m = af.AudioFrame.new([0.5, 0.25, -0.5, -0.25], 16000, 2, 1000).to_mono()   # capture.rs:385-400
    assert m.channels == 1 and m.samples.size == 2
    assert abs(m.samples[0] - 0.375) < 0.001 and abs(m.samples[1] + 0.375) < 0.001
    m1 = af.AudioFrame.new([0.5, -0.5], 16000, 1, 1000).to_mono()               # capture.rs:372-382
    assert bits_equal(m1.samples, np.array([0.5, -0.5], np.float32))


@pytest.mark.parametrize("channels", [1, 2, 3, 4, 6])
@pytest.mark.parametrize("n", [0, 1, 2, 5, 4096, 100003])
def test_to_mono_bit_exact(af, po, channels, n):
    x = np.random.default_rng(n + channels).uniform(-1, 1, n).astype(np.float32)
    got = af.to_mono(x, channels)
    want = po.to_mono(x, channels)
    assert got.size == want.size == (n if channels == 1 else -(-n // channels))
    assert bits_equal(got, want)


def _signed_zero_frames(channels: int, n_frames: int, tail: int = 0) -> np.ndarray:
    """Interleaved frames cycling through the signed-zero cases: all -0.0, all +0.0, -0.0 first then +0.0s, +0.0 first
    then -0.0s, a value among -0.0s, opposite values (sum +0.0), -0.0s behind a value; `tail` extra samples of -0.0
    (a trailing partial frame, `chunks` semantics)."""
    nz, pz = np.float32(-0.0), np.float32(0.0)
    pats = [[nz] * channels, [pz] * channels, [nz] + [pz] * (channels - 1), [pz] + [nz] * (channels - 1),
            [nz] * (channels - 1) + [np.float32(-1.5)], [np.float32(0.25), np.float32(-0.25)] + [nz] * (channels - 2),
            [np.float32(0.75)] + [nz] * (channels - 1)]
    x = np.array([pats[i % len(pats)] for i in range(n_frames)], np.float32).reshape(-1)
    return np.concatenate([x, np.full(tail, nz, np.float32)])


@pytest.mark.parametrize("channels,n_frames,tail", [(2, 7, 0), (2, 7, 1), (2, 20011, 0), (2, 20011, 1), (3, 7, 2), (3, 5000, 1),
                                                    (4, 7, 3), (4, 5000, 0), (6, 33, 5)])
def test_to_mono_signed_zero(af, po, channels, n_frames, tail):
    """docs/OPS.md "to_mono", the fold's seed: `chunk.iter().sum()` (capture.rs:36) folds from -0.0, so a frame of all
    -0.0 stays -0.0; to_mono2_kernel (stereo pairs), to_mono_kernel (everything else, incl. the tail of a stereo buffer)
    and the oracle give the same BITS on every signed-zero case."""
    x = _signed_zero_frames(channels, n_frames, tail)
    got, want = af.to_mono(x, channels), po.to_mono(x, channels)
    assert bits_equal(got, want)
    assert np.signbit(got[0]) and not np.signbit(got[1])            # all -0.0 -> -0.0, all +0.0 -> +0.0
    assert not np.signbit(got[2]) and not np.signbit(got[3])         # mixed zeros -> +0.0
    assert got[5] == 0.0 and not np.signbit(got[5])                  # x + (-x) = +0.0
    if tail:
        assert np.signbit(got[-1])                                   # the partial frame of -0.0s


def test_to_mono_signed_zero_device_both_kernels(af, po):
    """A stereo buffer long enough that the pair kernel takes the bulk and the general kernel the rest: same bits."""
    torch = pytest.importorskip("torch")
    import ctypes as C
    x = _signed_zero_frames(2, 300001, 1)
    dev = torch.device("cuda:0")
    xd = torch.from_numpy(x).to(dev)
    out = torch.empty((x.size + 1) // 2, device=dev)
    assert af.library().af_to_mono_device(C.c_void_p(xd.data_ptr()), x.size, 2, C.c_void_p(out.data_ptr()), None) == 0
    torch.cuda.synchronize()
    assert bits_equal(out.cpu().numpy(), po.to_mono(x, 2))


def _stereo_with_signed_zero_blocks(n_frames: int, seed: int) -> np.ndarray:
    """[n_frames, 2] f32: a synthetic clip with stretches of (-0.0, -0.0), (-0.0, +0.0), (+0.0, -0.0) and (x, -x) frames."""
    x = synth_clip(seed, n_frames, 2, 48000).reshape(n_frames, 2).copy()
    q = n_frames // 8
    x[q:q + 700] = np.float32(-0.0)
    x[2 * q:2 * q + 300, 0] = np.float32(-0.0); x[2 * q:2 * q + 300, 1] = np.float32(0.0)
    x[3 * q:3 * q + 300, 0] = np.float32(0.0); x[3 * q:3 * q + 300, 1] = np.float32(-0.0)
    x[4 * q:4 * q + 500, 1] = -x[4 * q:4 * q + 500, 0]
    x[-5:] = np.float32(-0.0)
    return x


def test_fused_downmix_equals_to_mono_on_signed_zeros(af, po):
    """The fused loads ((L + R) * 0.5 in the feature, resampler and chain kernels) against the stand-alone to_mono in
    front of the mono form of the same kernel, on input full of signed zeros: identical bits."""
    n = 48000
    x = _stereo_with_signed_zero_blocks(n, 3)
    mono = af.to_mono(x.reshape(-1), 2)
    assert bits_equal(mono, po.to_mono(x.reshape(-1), 2))
    q = n // 8
    assert np.signbit(mono[q:q + 700]).all() and not np.signbit(mono[2 * q:2 * q + 300]).any()
    # feature kernel (hop 256: features_v2; hop 160: the general kernel)
    for hop in (256, 160):
        f2 = af.FeaturePipeline(sample_rate=48000, channels=2, hop=hop)
        f1 = af.FeaturePipeline(sample_rate=48000, channels=1, hop=hop)
        f2.set_batch_uniform(1, n); f1.set_batch_uniform(1, n)
        assert bits_equal(f2.run(x), f1.run(mono))
    # resampler with stereo staging (one block matrix: 48 -> 16 kHz; ten: 44.1 -> 16 kHz), against to_mono + mono resampler
    for rin, rout in ((48000, 16000), (44100, 16000)):
        assert bits_equal(af.resample_stereo_batch(x.reshape(1, -1), rin, rout), af.resample_batch(mono[None, :], rin, rout))
    # chain: stereo in (downmix fused into the first launch) against mono in
    taps = np.hamming(129).astype(np.float32); taps /= taps.sum()
    c2 = af.ChainPipeline(input_rate=48000, channels=2, output_rate=16000, fir_taps=taps)
    c1 = af.ChainPipeline(input_rate=48000, channels=1, output_rate=16000, fir_taps=taps)
    c2.set_batch_uniform(1, n); c1.set_batch_uniform(1, n)
    assert bits_equal(c2.run(x), c1.run(mono))


def test_to_mono_full_size_round_trip(af, po):
    torch = pytest.importorskip("torch")
    import ctypes as C
    dev = torch.device("cuda:0")
    n_frames = 30 * 48000 * 16                    # 16 clips of config 2
    x = torch.rand(n_frames, 2, device=dev) * 2 - 1
    out = torch.empty(n_frames, device=dev)
    st = af.library().af_to_mono_device(C.c_void_p(x.data_ptr()), x.numel(), 2,
                                        C.c_void_p(out.data_ptr()), None)
    assert st == 0
    torch.cuda.synchronize()
    assert bits_equal(out, (x[:, 0] + x[:, 1]) / 2)           # exact: one add, one halving
    # to_mono of [m, m] interleaved is the identity (idempotence on duplicated channels)
    dup = torch.stack([out, out], dim=1).contiguous()
    out2 = torch.empty_like(out)
    assert af.library().af_to_mono_device(C.c_void_p(dup.data_ptr()), dup.numel(), 2,
                                          C.c_void_p(out2.data_ptr()), None) == 0
    torch.cuda.synchronize()
    assert bits_equal(out2, out)


# ---- quantiser + base64 ----------------------------------------------------------------
def test_quantiser_reference_vector_and_edges(af, po):
    assert af.quantize_pcm16([0.5, -0.5, 0.25, -0.25]).tolist() == [16383, -16383, 8191, -8191]
    x = np.array([1.0, -1.0, 2.0, -2.0, np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-9, 0.99999,
                  3.0517578e-05, -3.0517578e-05, 0.1, -0.1], np.float32)
    assert bits_equal(af.quantize_pcm16(x), po.quantize_pcm16(x))


@pytest.mark.parametrize("n", [0, 1, 3, 4, 5, 320, 1001, 250007])
def test_quantiser_bit_exact(af, po, n):
    x = (np.random.default_rng(n).standard_normal(n) * 0.6).astype(np.float32)
    assert bits_equal(af.quantize_pcm16(x), po.quantize_pcm16(x))


@pytest.mark.parametrize("n", [0, 1, 2, 3, 4, 5, 6, 7, 11, 12, 13, 320, 1000, 99991])
def test_pcm16_base64_exact(af, po, n):
    x = (np.random.default_rng(100 + n).standard_normal(n) * 0.5).astype(np.float32)
    got = af.MessageBuilder.audio_base64(x)
    want = base64.b64encode(po.quantize_pcm16(x).astype("<i2").tobytes()).decode()
    assert got == want
    assert len(got) == af.library().af_base64_len(2 * n)


def test_audio_message_envelope(af):
    import json
    msg = af.MessageBuilder.audio_message([0.5, -0.5, 0.25, -0.25])              # websocket.rs:376-382
    assert "audio_base_64" in msg and "input_audio_chunk" in msg
    d = json.loads(msg)
    raw = base64.b64decode(d["audio_base_64"])
    assert np.frombuffer(raw, "<i2").tolist() == [16383, -16383, 8191, -8191]


# ---- VAD -------------------------------------------------------------------------------
def test_energy_reference_vectors(af):
    assert af.vad_energy(np.zeros(480, np.float32), 480)[0] == 0.0               # vad.rs:288-291
    assert abs(af.vad_energy(np.full(480, 0.5, np.float32), 480)[0] - 0.25) < 0.0001   # :294-297


@pytest.mark.parametrize("frame_len", [1, 3, 63, 64, 160, 480, 1000, 1023])
def test_energy_is_the_reference_fold_bit_for_bit(af, po, frame_len):
    # vad.rs:163-165 folds left in f32; one lane per frame does the same on the device
    n_frames = 300                                                   # more than one wave's 64 lanes
    x = synth_clip(40, frame_len * n_frames, 1, 16000)
    got = af.vad_energy(x, frame_len)
    want = np.array([po.vad_energy(x[i * frame_len:(i + 1) * frame_len]) for i in range(n_frames)], np.float32)
    assert bits_equal(got, want)
    off = af.vad_energy(x[1:1 + frame_len * 7], frame_len)           # 4-byte aligned only: scalar loads
    assert bits_equal(off, np.array([po.vad_energy(x[1 + i * frame_len:1 + (i + 1) * frame_len])
                                         for i in range(7)], np.float32))


def test_vad_reference_state_tests(af):
    V, S, C = af.VoiceActivityDetector, af.VadState, af.VadConfig
    assert V(C(threshold_db=-50.0)).detect(np.full(480, 0.0001, np.float32)) == S.Silence  # vad.rs:211
    assert V(C(threshold_db=-50.0)).detect(np.full(480, 0.5, np.float32)) == S.Speech     # vad.rs:225
    v = V(C(threshold_db=-50.0, silence_timeout_frames=2, min_speech_frames=1,
            smoothing_factor=0.0))                                                       # vad.rs:239-265
    assert v.state() == S.Silence
    speech, silence = np.full(480, 0.5, np.float32), np.full(480, 0.0001, np.float32)
    assert v.detect(speech) == S.Speech
    assert v.detect(silence) == S.Speech
    assert v.detect(silence) == S.Ending
    assert v.detect(silence) == S.Silence
    v = V(C())                                                                           # vad.rs:267-281
    v.detect(speech)
    assert v.is_speaking()
    v.reset()
    assert v.state() == S.Silence and not v.is_speaking()


def test_vad_state_sequence_vs_oracle(af, po):
    # a long synthetic utterance: bursts of tone between near-silence, 20 ms frames @ 16 kHz
    rng = np.random.default_rng(5)
    frame_len, n_frames = 320, 400
    x = np.zeros(frame_len * n_frames, np.float32)
    t = np.arange(frame_len) / 16000.0
    for f in range(n_frames):
        amp = 0.3 if (f // 23) % 2 == 0 else 0.0005
        x[f * frame_len:(f + 1) * frame_len] = amp * np.sin(2 * np.pi * 220 * t) \
            + 1e-5 * rng.standard_normal(frame_len)
    v = af.VoiceActivityDetector(af.VadConfig())
    got = [int(s) for s in v.detect_frames(x, frame_len)]
    o = po.Vad()
    want = [o.detect(x[f * frame_len:(f + 1) * frame_len]) for f in range(n_frames)]
    assert got == want
    assert {0, 1, 2} <= set(got)                   # the sequence exercises all three states
    assert v.speech_frame_count() == o.speech_frames


def test_vad_batch_reference_state_tests(af):
    """The reference's state tests (vad.rs:211-281) through the batched device scan: one clip per test."""
    C = af.VadConfig
    speech, silence = np.full(480, 0.5, np.float32), np.full(480, 0.0001, np.float32)
    got = af.VoiceActivityDetector.detect_batch(np.stack([silence, speech]), 480, C(threshold_db=-50.0))
    assert got.tolist() == [[0], [1]]                                                    # vad.rs:211, :225
    seq = np.concatenate([speech, silence, silence, silence])[None, :]
    got = af.VoiceActivityDetector.detect_batch(seq, 480, C(threshold_db=-50.0, silence_timeout_frames=2,
                                                            min_speech_frames=1, smoothing_factor=0.0))
    assert got.tolist() == [[1, 1, 2, 0]]                                                # vad.rs:239-265


@pytest.mark.parametrize("cfg", [dict(), dict(smoothing_factor=0.0, silence_timeout_frames=2, min_speech_frames=1),
                                 dict(threshold_db=-30.0, smoothing_factor=0.9, silence_timeout_frames=4),
                                 dict(threshold_db=float("-inf")), dict(threshold_db=float("inf")),
                                 dict(threshold_db=-50.0, min_speech_frames=50)])
def test_vad_batch_device_scan_equals_oracle(af, po, cfg):
    """1024 clips x 1500 frames of energies around the threshold (incl. 0, denormals, inf, NaN): the states of
    the device scan equal n_clips oracle detectors exactly, and a scan cut in two pieces with the lane records
    carried over equals the one-piece scan."""
    import torch
    rng = np.random.default_rng(11)
    n_clips, n_frames = 1024, 1500
    thr = cfg.get("threshold_db", -50.0)
    centre = 10.0 ** (thr / 20.0) if np.isfinite(thr) else 1e-3
    e = (centre * np.exp(rng.normal(0, 0.02, (n_clips, n_frames)))).astype(np.float32)      # hugging the threshold
    burst = (np.arange(n_frames)[None, :] // rng.integers(5, 60, (n_clips, 1))) % 2 == 0
    e = np.where(burst, e * np.float32(3.0), e * np.float32(0.3)).astype(np.float32)
    e[0, :8] = [0.0, 1e-45, 1e-39, np.inf, np.nan, -1.0, 3.4e38, centre]
    want = po.vad_scan_batch(e, **cfg)
    dev = torch.device("cuda:0")
    ed = torch.from_numpy(e).to(dev)
    st = torch.zeros(n_clips, n_frames, dtype=torch.uint8, device=dev)
    c = af.VadConfig(**cfg)
    af.VoiceActivityDetector.detect_batch_device(ed.data_ptr(), n_clips, n_frames, st.data_ptr(), c)
    torch.cuda.synchronize()
    assert bits_equal(st.cpu().numpy(), want)
    if np.isfinite(thr):
        assert {0, 1} <= set(np.unique(want).tolist())
    # streaming: two pieces (the second one not a multiple of 4 frames in), lanes carried on the device
    lanes = torch.zeros(n_clips, 4, dtype=torch.int32, device=dev)
    st2 = torch.zeros_like(st)
    cut = 701
    a, b = ed[:, :cut].contiguous(), ed[:, cut:].contiguous()
    sa = torch.zeros(n_clips, cut, dtype=torch.uint8, device=dev)
    sb = torch.zeros(n_clips, n_frames - cut, dtype=torch.uint8, device=dev)
    af.VoiceActivityDetector.detect_batch_device(a.data_ptr(), n_clips, cut, sa.data_ptr(), c, lanes.data_ptr())
    af.VoiceActivityDetector.detect_batch_device(b.data_ptr(), n_clips, n_frames - cut, sb.data_ptr(), c,
                                                 lanes.data_ptr())
    torch.cuda.synchronize()
    assert bits_equal(torch.cat([sa, sb], 1).cpu().numpy(), want)


def test_vad_batch_host_entry_equals_single_handles(af, po):
    rng = np.random.default_rng(3)
    n_clips, n_frames, frame_len = 37, 61, 160
    x = (rng.standard_normal((n_clips, n_frames * frame_len)) * 0.01).astype(np.float32)
    x *= (rng.random((n_clips, n_frames)) < 0.5).repeat(frame_len, axis=1).astype(np.float32) * 30 + 0.05
    got = af.VoiceActivityDetector.detect_batch(x, frame_len)
    for c in range(0, n_clips, 6):
        v = af.VoiceActivityDetector(af.VadConfig())
        assert [int(s) for s in v.detect_frames(x[c], frame_len)] == got[c].tolist()


# ---- level meter (specs/0002-design.md:727-732) ------------------------------------------------
@pytest.mark.parametrize("window", [1, 64, 800, 2400])
def test_level_meter_vs_oracle(af, po, window):
    x = synth_clip(41, window * 23, 1, 16000)
    x[:window] = 0.0                                     # a silent window: dbfs = -inf
    got = af.level_meter(x, window)
    assert len(got) == 23
    for i, lv in enumerate(got):
        r, p, d = po.level_meter(x[i * window:(i + 1) * window])
        assert abs(lv.rms - r) <= 1e-6 * max(r, 1e-30)
        assert lv.peak == np.float32(p)                  # max |x| is exact
        if r == 0.0:
            assert lv.dbfs == float("-inf")
        else:
            assert abs(lv.dbfs - d) < 1e-4
