"""GPU parity of the wire path in one launch (af_wire_*): interleaved PCM -> to_mono (capture.rs:30-42) -> 48 -> 16 kHz
(resampler.rs:60-93; docs/OPS.md polyphase resampler) -> (x.clamp(-1, 1) * 32767) as i16 LE (websocket.rs:246-251) ->
padded base64 (websocket.rs:254), per clip or per message of `frame_samples` samples (specs/0002-design.md:1543-1546).

Bytes are compared
  * with the reference-pinned oracle codec (po.quantize_pcm16 + base64) applied to the f32 signal the product's own
    stand-alone resampler delivers (the fused launch computes that signal with the same block matrices in the same order),
  * with the existing entry points (MessageBuilder.audio_base64 of the resampled clip),
  * with the stage-by-stage form of the same handle ("unfused"),
all bit for bit; against the float64 oracle CHAIN (to_mono -> resample -> quantise) the decoded samples may differ by one
LSB where the f32 and f64 resampled values straddle an integer (tolerance stated in the test)."""
import base64

import numpy as np
import pytest

from util import bits_equal, synth_clip

pytestmark = pytest.mark.gpu

F32, I16 = 0, 1


def _b64(q: np.ndarray) -> bytes:
    return base64.b64encode(np.ascontiguousarray(q, "<i2").tobytes())


def _clips(n_clips, n_frames, channels, seed=0, gain=2.5):
    """[n_clips, n_frames * channels] f32; gain > 1 so that the clamp is exercised."""
    x = np.stack([synth_clip(seed + i, n_frames, channels, 48000) for i in range(n_clips)])
    return (x * np.float32(gain)).astype(np.float32)


def _mono_f32(x, channels, i16):
    """The mono f32 signal the fused staging computes: x / 32768 for i16, (L + R) * 0.5 for stereo (docs/OPS.md)."""
    v = x.astype(np.float32) * np.float32(1.0 / 32768.0) if i16 else x
    if channels == 2:
        v = v.reshape(v.shape[0], -1, 2)
        v = ((v[:, :, 0] + v[:, :, 1]) * np.float32(0.5)).astype(np.float32)
    return np.ascontiguousarray(v)


@pytest.mark.parametrize("n_frames", [30001, 30004, 30007, 11520, 11521, 3, 1])
def test_wire_per_clip_string_equals_the_separate_stages(af, po, n_frames):
    """stereo f32, 48 -> 16 kHz, output lengths 0, 1, 2 mod 3; one tile exactly, one sample more, tiny clips"""
    n_clips = 3
    x = _clips(n_clips, n_frames, 2)
    w = af.WireEncoder()
    assert w.n_launches() == 1
    total = w.set_batch_uniform(n_clips, n_frames)
    n_res = w.out_samples(n_frames)
    assert n_res == -(-n_frames // 3) and total == n_clips * 4 * (-(-2 * n_res // 3))
    text, q = w.run(x, with_pcm16=True)
    y = af.resample_stereo_batch(x, 48000, 16000)                  # the product's stand-alone resampler, f32
    tl = total // n_clips
    for i in range(n_clips):
        want_q = po.quantize_pcm16(y[i])                           # the reference's quantiser (oracle) on that signal
        assert bits_equal(q[i * n_res:(i + 1) * n_res], want_q)
        assert text[i * tl:(i + 1) * tl] == _b64(want_q)
        assert text[i * tl:(i + 1) * tl].decode() == af.MessageBuilder.audio_base64(y[i])    # the existing entry point
    w.set_option("unfused", 1)
    assert w.n_launches() == 2
    text2, q2 = w.run(x, with_pcm16=True)
    assert text2 == text and bits_equal(q2, q)


def test_wire_nan_inf_and_clamp(af, po):
    n = 24000
    x = _clips(2, n, 2, seed=5, gain=1.0).reshape(2, n, 2)
    x[0, 1000, 0] = np.nan
    x[0, 5000:5003, 1] = np.inf
    x[0, 9000, 0] = -np.inf
    x[1, 300:400] *= np.float32(40.0)                              # far beyond +-1
    x[1, 7000, :] = np.float32(-0.0)
    x = x.reshape(2, -1)
    w = af.WireEncoder()
    total = w.set_batch_uniform(2, n)
    text, q = w.run(x, with_pcm16=True)
    y = af.resample_stereo_batch(x, 48000, 16000)
    assert np.isnan(y[0]).any() and (np.abs(y[1]) > 1.5).any()
    want = np.concatenate([po.quantize_pcm16(y[i]) for i in range(2)])
    assert bits_equal(q, want)                                     # NaN -> 0, +-inf and |x| > 1 -> +-32767
    assert (q[:8000][np.isnan(y[0])] == 0).all() and q.max() == 32767 and q.min() == -32767
    assert text == b"".join(_b64(po.quantize_pcm16(y[i])) for i in range(2))
    w.set_option("unfused", 1)
    assert w.run(x) == text


@pytest.mark.parametrize("channels,fmt", [(1, F32), (2, I16), (1, I16), (2, F32)])
@pytest.mark.parametrize("rates", [(48000, 16000), (32000, 16000), (48000, 24000)])
def test_wire_formats_and_single_block_ratios(af, po, channels, fmt, rates):
    n_frames, n_clips = (20011 if not (channels == 1 and fmt == I16) else 20012), 2     # (mono i16 clips start on even frames)
    xf = _clips(n_clips, n_frames, channels, seed=11, gain=1.7)
    x = np.clip(np.round(xf * 20000.0), -32768, 32767).astype(np.int16) if fmt == I16 else xf
    w = af.WireEncoder(input_rate=rates[0], output_rate=rates[1], channels=channels, sample_format=fmt)
    assert w.n_launches() == 1
    w.set_batch_uniform(n_clips, n_frames)
    text, q = w.run(x, with_pcm16=True)
    mono = _mono_f32(x, channels, fmt == I16)
    y = af.resample_batch(mono, rates[0], rates[1])
    want = [po.quantize_pcm16(y[i]) for i in range(n_clips)]
    assert bits_equal(q, np.concatenate(want))
    assert text == b"".join(_b64(v) for v in want)


@pytest.mark.parametrize("frame", [320, 160, 480, 3840, 2])
def test_wire_messages(af, po, frame):
    """one padded base64 string per `frame` resampled samples; the trailing partial batch is not sent"""
    n_frames, n_clips = 48000 + 957, 2                              # 16 319 resampled samples
    x = _clips(n_clips, n_frames, 2, seed=21)
    w = af.WireEncoder(frame_samples=frame)
    assert w.n_launches() == 1
    w.set_batch_uniform(n_clips, n_frames)
    n_res = w.out_samples(n_frames)
    assert w.n_messages(n_frames) == n_res // frame and w.message_len() == 4 * (-(-2 * frame // 3))
    msgs = w.messages(x)
    y = af.resample_stereo_batch(x, 48000, 16000)
    for i in range(n_clips):
        assert len(msgs[i]) == n_res // frame
        for k in (0, 1, len(msgs[i]) // 2, len(msgs[i]) - 1):
            assert msgs[i][k] == af.MessageBuilder.audio_base64(y[i, k * frame:(k + 1) * frame])    # websocket.rs:246-254
            assert msgs[i][k].encode() == _b64(po.quantize_pcm16(y[i, k * frame:(k + 1) * frame]))
    whole, q = w.run(x, with_pcm16=True)
    # the PCM16 rows hold every resampled sample, those of the unsent trailing batch included
    assert bits_equal(q, np.concatenate([po.quantize_pcm16(y[i]) for i in range(n_clips)]))
    # (device buffers poisoned first: a sample the launch does not write would show)
    import torch
    xd = torch.from_numpy(x).cuda()
    for unfused in (0, 1):
        w.set_option("unfused", unfused)
        td = torch.full((w.total_text,), 0x23, dtype=torch.uint8, device="cuda")
        qd = torch.full((w.total_samples,), -21846, dtype=torch.int16, device="cuda")
        w.run_device(xd.data_ptr(), td.data_ptr(), qd.data_ptr())
        torch.cuda.synchronize()
        assert td.cpu().numpy().tobytes() == whole and bits_equal(qd.cpu().numpy(), q)


@pytest.mark.parametrize("frame", [320, 160, 480, 640, 8, 96])
def test_wire_message_energy_and_peak(af, po, frame):
    """Per message, beside the text: the mean-square energy VoiceActivityDetector::calculate_energy computes on the
    message (vad.rs:157-168 -- an f32 LEFT fold: bits equal to the oracle's and to af_vad_energy on the stand-alone
    resampler's signal) and max |x| (the audio_level event's peak, events/mod.rs:73-75).  In the fused launch the 16 kHz
    f32 signal exists in LDS only, so this is where a VAD gets its input; the text does not change."""
    n_frames, n_clips = 48000 * 2 + 1234, 3
    x = _clips(n_clips, n_frames, 2, seed=61, gain=1.3)
    x[1, :9000] = 0.0                                               # silence: energy exactly 0
    w = af.WireEncoder(frame_samples=frame)
    w.set_batch_uniform(n_clips, n_frames)
    plain = w.run(x)
    text, en, pk = w.run(x, with_level=True)
    assert text == plain
    y = af.resample_stereo_batch(x, 48000, 16000)
    nm = w.n_messages(n_frames)
    assert en.size == pk.size == n_clips * nm == w.total_messages()
    for i in range(n_clips):
        msgs = y[i, :nm * frame].reshape(nm, frame)
        assert bits_equal(en[i * nm:(i + 1) * nm], af.vad_energy(msgs.reshape(-1), frame))          # the stand-alone kernel
        want = np.array([po.vad_energy(m) for m in msgs[:40]], np.float32)                          # the oracle's left fold
        assert bits_equal(en[i * nm:i * nm + want.size], want)
        assert bits_equal(pk[i * nm:(i + 1) * nm], np.abs(msgs).max(axis=1))
    assert (en[nm:nm + 9000 // 2 // 3 // frame - 1] == 0.0).all()       # (9 000 interleaved samples = 4 500 frames = 1 500 resampled)
    w.set_option("unfused", 1)
    text2, en2, pk2 = w.run(x, with_level=True)
    assert text2 == text and bits_equal(en2, en) and bits_equal(pk2, pk)
    if frame % 80:
        return                                                      # (such messages take the stage-by-stage form: equal-length batches)
    # ragged batch: the clips' messages back to back
    lens = [7000, 48000, 5, 33333]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    xr = _clips(1, int(off[-1]), 2, seed=62)[0]
    w2 = af.WireEncoder(frame_samples=frame)
    w2.set_batch(off)
    _, en_r, pk_r = w2.run(xr, with_level=True)
    k = 0
    for i, n in enumerate(lens):
        yi = af.resample_stereo_batch(xr.reshape(-1, 2)[off[i]:off[i + 1]].reshape(1, -1), 48000, 16000)[0]
        m = yi.size // frame
        if m:
            assert bits_equal(en_r[k:k + m], af.vad_energy(yi[:m * frame], frame))
            assert bits_equal(pk_r[k:k + m], np.abs(yi[:m * frame].reshape(m, frame)).max(axis=1))
        k += m
    assert k == en_r.size


def test_wire_energy_feeds_the_batched_vad(af, po):
    """the energies of the fused launch through the reference's state machine (vad.rs:97-151, af_vad_detect_batch on the
    stand-alone resampler's signal as the comparison): identical states"""
    n_frames = 48000 * 4
    x = _clips(2, n_frames, 2, seed=71, gain=1.0).reshape(2, n_frames, 2)
    x[0, 40000:90000] *= np.float32(1e-4)                            # a stretch below the threshold
    x[1, :60000] = 0.0
    x = x.reshape(2, -1)
    w = af.WireEncoder(frame_samples=320)
    w.set_batch_uniform(2, n_frames)
    _, en, _ = w.run(x, with_level=True)
    y = af.resample_stereo_batch(x, 48000, 16000)
    nm = w.n_messages(n_frames)
    want = af.VoiceActivityDetector.detect_batch(y[:, :nm * 320], 320)
    got = po.vad_scan_batch(en.reshape(2, nm))
    assert np.array_equal(np.asarray(want).reshape(2, nm), np.asarray(got).reshape(2, nm))
    assert len(set(np.asarray(got).reshape(-1).tolist())) >= 2      # both silence and speech occur


def test_wire_frame_outside_the_fused_tile_runs_the_stages(af, po):
    """a message length that does not divide the tile (300) / a ratio with ten block matrices (44.1 k): two launches"""
    x = _clips(2, 30000, 2, seed=31)
    w = af.WireEncoder(frame_samples=300)
    assert w.n_launches() == 2
    w.set_batch_uniform(2, 30000)
    msgs = w.messages(x)
    y = af.resample_stereo_batch(x, 48000, 16000)
    assert len(msgs[1]) == 10000 // 300
    assert msgs[1][7].encode() == _b64(po.quantize_pcm16(y[1, 2100:2400]))
    w2 = af.WireEncoder(input_rate=44100, frame_samples=0)
    assert w2.n_launches() == 2
    w2.set_batch_uniform(2, 30000)
    y2 = af.resample_stereo_batch(x, 44100, 16000)
    assert w2.run(x) == b"".join(_b64(po.quantize_pcm16(y2[i])) for i in range(2))
    with pytest.raises(af.AudioError):
        af.WireEncoder(input_rate=44100, sample_format=I16)        # i16 has no stage-by-stage form


def test_wire_mono_i16_alignment_is_checked(af):
    """mono i16 is staged one dword per frame pair: clips on odd frames, or a PCM pointer that is only 2-byte aligned, are
    refused (AudioError), never read across the end of the buffer"""
    torch = pytest.importorskip("torch")
    w = af.WireEncoder(channels=1, sample_format=I16)
    with pytest.raises(af.AudioError):
        w.set_batch_uniform(2, 20011)
    with pytest.raises(af.AudioError):
        w.set_batch(np.array([0, 7001, 9000], np.int64))
    w.set_batch_uniform(1, 20011)                                   # one clip of an odd count is fine
    x = torch.zeros(20011 + 8, dtype=torch.int16, device="cuda:0")
    text = torch.empty(w.total_text, dtype=torch.uint8, device="cuda:0")
    w.run_device(x.data_ptr(), text.data_ptr())
    with pytest.raises(af.AudioError):
        w.run_device(x.data_ptr() + 2, text.data_ptr())


def test_wire_ragged_batch(af, po):
    lens = [0, 1, 5, 7000, 48000, 33333, 11520 * 3, 2]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    x = _clips(1, int(off[-1]), 2, seed=41)[0].reshape(-1, 2)
    for frame in (0, 320):
        w = af.WireEncoder(frame_samples=frame)
        w.set_batch(off)
        text, q = w.run(x.reshape(-1), with_pcm16=True)
        toff = w.text_offsets()
        qo = 0
        for i, n in enumerate(lens):
            n_res = -(-n // 3)
            y = af.resample_stereo_batch(x[off[i]:off[i + 1]].reshape(1, -1), 48000, 16000)[0] if n else np.zeros(0, np.float32)
            want_q = po.quantize_pcm16(y)
            assert bits_equal(q[qo:qo + n_res], want_q)
            qo += n_res
            if frame:
                want = b"".join(_b64(want_q[k * frame:(k + 1) * frame]) for k in range(n_res // frame))
            else:
                want = _b64(want_q)
            assert text[toff[i]:toff[i + 1]] == want, (frame, i)


def test_wire_against_the_float64_oracle_chain(af, po):
    """to_mono -> resample (float64) -> quantise in the oracle: the decoded samples agree within one LSB (the f32 and the
    f64 resampled value can lie on either side of an integer after * 32767), and exactly in all but a small fraction"""
    n = 48000
    x = _clips(2, n, 2, seed=51, gain=0.9)
    w = af.WireEncoder()
    w.set_batch_uniform(2, n)
    text = w.run(x)
    tl = len(text) // 2
    for i in range(2):
        got = np.frombuffer(base64.b64decode(text[i * tl:(i + 1) * tl]), "<i2").astype(np.int32)
        ref = po.quantize_pcm16(po.resample(po.to_mono(x[i], 2), 48000, 16000).astype(np.float32)).astype(np.int32)
        d = np.abs(got - ref)
        assert d.max() <= 1 and (d != 0).mean() < 2e-3


def test_wire_full_size_batch_fused_equals_stages(af):
    """64 clips x 30 s stereo 48 kHz on the device: the fused launch and the two-launch form give the same characters and
    samples; every clip of identical input gives identical text (size-independent properties)"""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    n_clips, n = 64, 1440000
    x = (torch.rand(n_clips, n, 2, device=dev) * 2 - 1) * 1.2
    x[7] = x[3]
    w = af.WireEncoder(frame_samples=320)
    total = w.set_batch_uniform(n_clips, n)
    text = torch.empty(total, dtype=torch.uint8, device=dev)
    q = torch.empty(w.total_samples, dtype=torch.int16, device=dev)
    w.run_device(x.data_ptr(), text.data_ptr(), q.data_ptr())
    torch.cuda.synchronize()
    text2, q2 = torch.empty_like(text), torch.empty_like(q)
    w.set_option("unfused", 1)
    w.run_device(x.data_ptr(), text2.data_ptr(), q2.data_ptr())
    torch.cuda.synchronize()
    assert bits_equal(text, text2) and bits_equal(q, q2)
    t = text.view(n_clips, -1)
    assert bits_equal(t[7], t[3]) and not bits_equal(t[7], t[4])
    # every message is 856 characters of the base64 alphabet ending in the two '=' of 640 = 213 * 3 + 1 bytes
    m = t[5].view(-1, 856).cpu().numpy()
    assert (m[:, -2:] == ord("=")).all() and (m[:, :-2] != ord("=")).all()
