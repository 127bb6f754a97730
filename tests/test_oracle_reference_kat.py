"""Pins the oracle's [R] restatements with the reference's OWN unit-test vectors.

Each test is the 1:1 port of a `#[test]` in forfd8960/audio-flow-rs
(src-tauri/src/...), cited per test.  This is the complete list of reference
tests that touch hot-path numerics.
"""
import base64
import json
import threading

import numpy as np

from util import bits_equal
import pytest


# ---- capture.rs: AudioFrame::to_mono -------------------------------------------
def test_audio_frame_to_mono_single_channel(po):
    # capture.rs:371-382
    assert bits_equal(po.to_mono([0.5, -0.5], 1), np.array([0.5, -0.5], np.float32))


def test_audio_frame_to_mono_stereo(po):
    # capture.rs:384-400: [L, R, L, R] -> (0.5 + 0.25)/2, (-0.5 + -0.25)/2
    mono = po.to_mono([0.5, 0.25, -0.5, -0.25], 2)
    assert mono.size == 2
    assert abs(mono[0] - 0.375) < 0.001
    assert abs(mono[1] - (-0.375)) < 0.001
    assert bits_equal(mono, np.array([0.375, -0.375], np.float32))     # and exactly so


def test_to_mono_partial_trailing_frame(po):
    # capture.rs:35-38 `.chunks(channels)`: the short last chunk is still divided by `channels`
    mono = po.to_mono([0.3, 0.1, 0.5], 2)
    assert mono.size == 2
    assert mono[0] == np.float32((np.float32(0.3) + np.float32(0.1)) / np.float32(2))
    assert mono[1] == np.float32(np.float32(0.5) / np.float32(2))


def test_to_mono_three_channels_sequential_f32_sum(po):
    x = np.array([0.1, 0.2, 0.3, 1e-8, 1.0, -1.0], np.float32)
    want = [np.float32((np.float32(np.float32(0.1) + np.float32(0.2)) + np.float32(0.3)) / np.float32(3)),
            np.float32((np.float32(np.float32(1e-8) + np.float32(1.0)) + np.float32(-1.0)) / np.float32(3))]
    assert bits_equal(po.to_mono(x, 3), np.array(want, np.float32))


# ---- capture.rs: RingBuffer ------------------------------------------------------
def test_ring_buffer_write_read(po):
    r = po.Ring(1024)                                   # capture.rs:432-442
    assert r.write([1.0, 2.0, 3.0]) == 3
    assert bits_equal(r.read(3), [1.0, 2.0, 3.0])


def test_ring_buffer_partial_read(po):
    r = po.Ring(1024)                                   # capture.rs:444-457
    r.write([1.0, 2.0, 3.0, 4.0, 5.0])
    assert bits_equal(r.read(2), [1.0, 2.0])
    assert bits_equal(r.read(3), [3.0, 4.0, 5.0])


def test_ring_buffer_wrap_around(po):
    r = po.Ring(10)                                     # capture.rs:459-471
    r.write([1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0])
    r.read(5)
    r.write([8.0, 9.0, 10.0])
    assert bits_equal(r.read(5), [6.0, 7.0, 8.0, 9.0, 10.0])


def test_ring_buffer_overflow(po):
    r = po.Ring(10)                                     # capture.rs:473-485
    assert r.write([1.0] * 20) == 9                     # one slot is reserved
    got = r.read(9)
    assert got.size == 9 and np.all(got == 1.0)


def test_ring_buffer_read_empty(po):
    assert po.Ring(1024).read(100) is None              # capture.rs:487-492


def test_ring_buffer_available(po):
    r = po.Ring(1024)                                   # capture.rs:494-504
    assert r.available() == 0
    r.write([1.0] * 100)
    assert r.available() == 100
    r.read(50)
    assert r.available() == 50


def test_ring_buffer_clear(po):
    r = po.Ring(1024)                                   # capture.rs:506-514
    r.write([1.0] * 100)
    assert r.available() == 100
    r.clear()
    assert r.available() == 0


def test_ring_buffer_cross_thread(po):
    r = po.Ring(1024)                                   # capture.rs:546-561
    t = threading.Thread(target=lambda: r.write([1.0, 2.0, 3.0]))
    t.start()
    t.join()
    assert bits_equal(r.read(3), [1.0, 2.0, 3.0])


# ---- resampler.rs ------------------------------------------------------------------
def test_chunker_bookkeeping(po):
    # resampler.rs:132-147: while buffer.len() >= 128 { process; drain(..128) }
    assert po.chunker_process(0, 100, 128) == (0, 100)
    assert po.chunker_process(100, 28, 128) == (1, 0)
    assert po.chunker_process(100, 300, 128) == (3, 16)
    assert po.chunker_process(127, 1, 128) == (1, 0)
    # equal rates => chunk_size 0 (resampler.rs:34-39): the reference would spin; oracle: passthrough
    assert po.chunker_process(0, 50, 0) == (0, 0)


# ---- vad.rs ------------------------------------------------------------------------
def test_energy_calculation(po):
    # vad.rs:283-298: mean-square, not RMS
    assert po.vad_energy(np.zeros(480, np.float32)) == 0.0
    assert abs(po.vad_energy(np.full(480, 0.5, np.float32)) - 0.25) < 0.0001
    assert po.vad_energy([]) == 0.0                     # vad.rs:158-160


def test_vad_silence_detection(po):
    v = po.Vad(threshold_db=-50.0)                      # vad.rs:211-223
    assert v.detect(np.full(480, 0.0001, np.float32)) == po.SILENCE


def test_vad_speech_detection(po):
    v = po.Vad(threshold_db=-50.0)                      # vad.rs:225-237
    assert v.detect(np.full(480, 0.5, np.float32)) == po.SPEECH


def test_vad_state_transitions(po):
    # vad.rs:239-265
    v = po.Vad(threshold_db=-50.0, silence_timeout_frames=2, min_speech_frames=1,
               smoothing_factor=0.0)
    assert v.state == po.SILENCE
    speech = np.full(480, 0.5, np.float32)
    silence = np.full(480, 0.0001, np.float32)
    assert v.detect(speech) == po.SPEECH
    assert v.detect(silence) == po.SPEECH
    assert v.detect(silence) == po.ENDING
    assert v.detect(silence) == po.SILENCE


def test_vad_reset(po):
    v = po.Vad()                                        # vad.rs:267-281
    v.detect(np.full(480, 0.5, np.float32))
    assert v.state == po.SPEECH
    v.reset()
    assert v.state == po.SILENCE


def test_vad_short_speech_is_dropped(po):
    # vad.rs:136-141: fewer than min_speech_frames => back to Silence without Ending
    v = po.Vad(silence_timeout_frames=2, min_speech_frames=3, smoothing_factor=0.0)
    speech = np.full(480, 0.5, np.float32)
    silence = np.zeros(480, np.float32)
    assert v.detect(speech) == po.SPEECH
    assert v.detect(silence) == po.SPEECH
    assert v.detect(silence) == po.SILENCE


def test_dbfs(po):
    assert po.energy_to_dbfs(0.0) == float("-inf")      # vad.rs:172-174
    assert abs(po.energy_to_dbfs(0.25) - 20.0 * np.log10(0.25)) < 1e-5


# ---- websocket.rs: quantiser + base64 ------------------------------------------------
def test_message_builder_audio(po):
    # websocket.rs:376-382 uses [0.5, -0.5, 0.25, -0.25]; :342 `(x.clamp(-1,1) * 32767.0) as i16`
    q = po.quantize_pcm16([0.5, -0.5, 0.25, -0.25])
    assert q.tolist() == [16383, -16383, 8191, -8191]   # truncation toward zero
    b64 = po.base64_encode(q.astype("<i2").tobytes())
    assert b64 == base64.b64encode(q.astype("<i2").tobytes()).decode()
    msg = json.dumps({"audio_base_64": b64, "message_type": "input_audio_chunk"})
    assert "audio_base_64" in msg and "input_audio_chunk" in msg


def test_quantiser_edges(po):
    x = np.array([1.0, -1.0, 2.0, -2.0, np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-9, 0.99999,
                  3.0517578e-05, -3.0517578e-05], np.float32)
    q = po.quantize_pcm16(x)
    assert q.tolist() == [32767, -32767, 32767, -32767, 0, 32767, -32767, 0, 0, 0, 32766, 0, 0]


@pytest.mark.parametrize("n", [0, 1, 2, 3, 4, 5, 6, 7, 100, 101, 102])
def test_base64_matches_rfc4648(po, n):
    data = bytes((i * 37 + 11) % 256 for i in range(n))
    assert po.base64_encode(data) == base64.b64encode(data).decode()
