"""Host-side mirror logic that needs no device: RingBuffer bookkeeping
(capture.rs:84-161, reference tests capture.rs:426-561), AudioError rendering
(error.rs:96-111), AudioFrame container, message envelopes."""
import json
import threading

import numpy as np

from util import bits_equal


def test_ring_buffer_reference_suite(af):
    r = af.RingBuffer(1024)
    assert r.capacity == 1024                                   # capture.rs:426-430
    assert r.write([1.0, 2.0, 3.0]) == 3                        # :432-442
    assert bits_equal(r.read(3), [1.0, 2.0, 3.0])
    r.write([1.0, 2.0, 3.0, 4.0, 5.0])                          # :444-457
    assert bits_equal(r.read(2), [1.0, 2.0])
    assert bits_equal(r.read(3), [3.0, 4.0, 5.0])
    assert r.read(100) is None                                  # :487-492

    r = af.RingBuffer(10)                                       # :459-471 wrap-around
    r.write([1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0])
    r.read(5)
    r.write([8.0, 9.0, 10.0])
    assert bits_equal(r.read(5), [6.0, 7.0, 8.0, 9.0, 10.0])

    r = af.RingBuffer(10)                                       # :473-485 overflow
    assert r.write([1.0] * 20) == 9
    assert bits_equal(r.read(9), [1.0] * 9)

    r = af.RingBuffer(1024)                                     # :494-514
    assert r.available() == 0
    r.write([1.0] * 100)
    assert r.available() == 100
    r.read(50)
    assert r.available() == 50
    r.clear()
    assert r.available() == 0

    r = af.RingBuffer(1024)                                     # :546-561
    t = threading.Thread(target=lambda: r.write([1.0, 2.0, 3.0]))
    t.start()
    t.join()
    assert bits_equal(r.read(3), [1.0, 2.0, 3.0])


def test_ring_buffer_matches_oracle_on_random_traffic(af, po):
    rng = np.random.default_rng(3)
    a, b = af.RingBuffer(37), po.Ring(37)
    for _ in range(400):
        if rng.random() < 0.55:
            d = rng.standard_normal(int(rng.integers(0, 30))).astype(np.float32)
            assert a.write(d) == b.write(d)
        else:
            n = int(rng.integers(0, 30))
            x, y = a.read(n), b.read(n)
            assert (x is None) == (y is None)
            if x is not None:
                assert bits_equal(x, y)
        assert a.available() == b.available()


def test_ring_buffer_c_abi_directly(af):
    """af_ring_* without the Python class: `None` vs `Some(vec![])` (capture.rs:131-135), capacity - 1
    usable slots (:111), truncated writes, a read into too small a buffer, a writer thread (:546-561)."""
    import ctypes as C
    import threading
    lib = af.library()
    h = C.c_void_p()
    assert lib.af_ring_create(10, C.byref(h)) == 0 and lib.af_ring_capacity(h) == 10
    out = np.zeros(16, np.float32)
    n, none = C.c_size_t(), C.c_int()
    assert lib.af_ring_read(h, 4, out.ctypes.data, out.size, C.byref(n), C.byref(none)) == 0
    assert (n.value, none.value) == (0, 1)                              # empty -> None
    x = np.arange(1, 16, dtype=np.float32)
    assert lib.af_ring_write(h, x.ctypes.data, x.size) == 9            # capacity - 1
    assert lib.af_ring_available(h) == 9
    assert lib.af_ring_read(h, 0, out.ctypes.data, out.size, C.byref(n), C.byref(none)) == 0
    assert (n.value, none.value) == (0, 0)                              # Some(vec![])
    assert lib.af_ring_read(h, 5, out.ctypes.data, 3, C.byref(n), None) == af._ffi.AF_ERR_BUFFER_TOO_SMALL
    assert lib.af_ring_available(h) == 9                                # nothing consumed by the failed read
    assert lib.af_ring_read(h, 5, out.ctypes.data, out.size, C.byref(n), None) == 0 and n.value == 5
    assert bits_equal(out[:5], x[:5])
    assert lib.af_ring_write(h, x.ctypes.data, 7) == 5                  # wraps, truncated to the free slots
    assert lib.af_ring_read(h, 100, out.ctypes.data, out.size, C.byref(n), None) == 0 and n.value == 9
    assert bits_equal(out[:9], np.concatenate([x[5:9], x[:5]]))
    lib.af_ring_clear(h)
    assert lib.af_ring_available(h) == 0
    t = threading.Thread(target=lambda: lib.af_ring_write(h, x.ctypes.data, 3))
    t.start(); t.join()
    assert lib.af_ring_available(h) == 3
    lib.af_ring_destroy(h)
    z = C.c_void_p()
    assert lib.af_ring_create(0, C.byref(z)) == 0                       # the reference would divide by zero
    assert lib.af_ring_write(z, x.ctypes.data, 3) == 0 and lib.af_ring_available(z) == 0
    lib.af_ring_destroy(z)


def test_audio_error_rendering(af):
    e = af.AudioError.resampling_failed("boom")
    assert str(e) == "Resampling failed: boom"                  # error.rs:109
    assert str(af.AudioError("NoDevice")) == "No input device available"   # error.rs:97
    assert af.AudioError("CaptureFailed", "x") == af.AudioError("CaptureFailed", "x")


def test_audio_frame_container(af):
    f = af.AudioFrame.new([0.5, -0.5, 0.25], 16000, 1, 1000)    # capture.rs:357-369
    assert f.samples.size == 3 and f.sample_rate == 16000 and f.channels == 1
    assert f.timestamp_ns == 1000
    m = f.to_mono()                                             # capture.rs:371-382 (1 ch: clone)
    assert m.channels == 1 and bits_equal(m.samples, f.samples)
    c = f.clone()                                               # capture.rs:536-544
    assert bits_equal(c.samples, f.samples) and c is not f


def test_configure_message(af):
    msg = af.MessageBuilder.configure_message("model_id", "en")   # websocket.rs:384-390
    d = json.loads(msg)
    assert d == {"model_id": "model_id", "language_code": "en", "encoding": "pcm_16000",
                 "message_type": "configure"}


def test_vad_config_defaults(af):
    c = af.VadConfig()                                          # vad.rs:34-43
    assert (c.threshold_db, c.smoothing_factor, c.silence_timeout_frames, c.min_speech_frames) \
        == (-50.0, 0.3, 15, 3)
    assert af.VadState.Silence == 0 and af.VadState.Speech == 1 and af.VadState.Ending == 2


def test_banded_plan_invariants(af):
    """Host-side shape selection of the MFMA resampler (banded_mfma.hip make_plan): no GPU."""
    import ctypes as C
    from math import gcd
    lib = af.library()
    names = ["ok", "L", "M", "nb", "S", "NK", "G", "wpp", "nwaves", "seg_len", "seg_stride", "smem"]

    def plan(fi, fo, ch=1):
        info = (C.c_int32 * 12)()
        ok = lib.af_resample_plan_describe(fi, fo, ch, info)
        d = dict(zip(names, list(info)))
        assert ok == d["ok"]
        return d

    # BASELINE's ratios
    p = plan(44100, 16000)
    assert (p["ok"], p["L"], p["M"], p["nb"], p["S"], p["NK"], p["G"], p["nwaves"]) == (1, 160, 441, 10, 441, 28, 2, 10)
    p = plan(48000, 16000)
    assert (p["ok"], p["L"], p["M"], p["nb"], p["S"], p["NK"], p["nwaves"]) == (1, 1, 3, 1, 48, 28, 8)
    assert plan(48000, 16000, 2)["ok"] == 1
    p2 = plan(44100, 16000, 2)                                # stereo staging of the ten-wave shape: same tiles as mono
    assert (p2["ok"], p2["nb"], p2["NK"], p2["G"], p2["nwaves"]) == (1, 10, 28, 2, 10)
    # more than 16 block matrices: the streamed form (G = 0 marks it): 8 waves, matrices fetched per tile, a tile =
    # a range of matrices x 32 block periods, one LDS segment (<= 511 samples, odd stride) per period
    for fi, fo, nb, S, nk in [(48000, 44100, 147, 2560, 24), (16000, 44100, 441, 2560, 20), (48000, 22050, 147, 5120, 28)]:
        p = plan(fi, fo)
        assert (p["ok"], p["nb"], p["S"], p["NK"], p["G"], p["nwaves"]) == (1, nb, S, nk, 0, 8)
        assert p["seg_len"] <= 511 and p["seg_stride"] % 2 == 1 and p["seg_stride"] >= p["seg_len"] + 2
        # (segments, plus the image of the tile's outputs where the deferred stores fit: two workgroups per CU either way)
        assert 32 * p["seg_stride"] * 4 <= p["smem"] <= 80 * 1024
        assert plan(fi, fo, 2)["ok"] == 0                     # stereo staging: resident matrices only
    short = 0
    for fi, fo in [(44100, 16000), (48000, 16000), (16000, 48000), (8000, 16000), (22050, 16000),
                   (44100, 48000), (32000, 16000), (96000, 16000), (11025, 16000), (16000, 8000),
                   (48000, 45000), (16000, 15000), (8000, 15000), (24000, 45000)]:
        for ch in (1, 2):
            p = plan(fi, fo, ch)
            if not p["ok"] or p["G"] == 0:                    # not covered / the streamed form (checked above)
                continue
            g = gcd(fi, fo)
            L, M = fo // g, fi // g
            assert (p["L"], p["M"]) == (L, M)
            assert p["nb"] == L // gcd(16, L) and p["S"] == 16 // gcd(16, L) * M
            assert p["nwaves"] == p["nb"] * p["wpp"] <= 16
            assert p["G"] % (2 * p["wpp"]) == 0               # chains run in pairs on every wave
            W = 64 + max(((16 * b * M) % L + 15 * M) // L for b in range(p["nb"]))
            assert 4 * p["NK"] >= W                           # the window fits the MFMA steps
            qp_last = 16 * (p["nb"] - 1) * M // L
            assert p["seg_len"] == (p["G"] - 1) * p["S"] + qp_last + 4 * p["NK"] <= min(1000, 2 * 64 * p["nwaves"])
            assert p["seg_stride"] % 2 == 1 and p["seg_stride"] >= p["seg_len"]
            assert p["smem"] == 16 * p["seg_stride"] * 4 <= 64 * 1024
            # segments may be shorter than the workgroup (the kernel masks the lanes beyond them)
            short += p["seg_len"] < 64 * p["nwaves"]
    assert short >= 2                                          # the L = 15 ratios exercise that mask


def test_chain_fused_plan_invariants(af):
    """Host-side shape of the chain's fused resampler + FIR launch (banded_mfma.hip make_fused_plan): no GPU."""
    import ctypes as C
    lib = af.library()
    names = ["ok", "NKA", "NKB", "H2", "pad", "P2", "seg_len_b", "seg_stride_b", "z_per_tile", "smem"]

    def plan(fi, fo, ch, K):
        info = (C.c_int32 * 10)()
        ok = lib.af_chain_fused_plan_describe(fi, fo, ch, K, info)
        d = dict(zip(names, list(info)))
        assert ok == d["ok"]
        return d

    # BASELINE config 4: stereo 48 kHz -> 16 kHz, 129 taps
    p = plan(48000, 16000, 2, 129)
    assert (p["ok"], p["NKA"], p["NKB"], p["H2"], p["pad"], p["P2"], p["z_per_tile"]) == (1, 28, 36, 128, 0, 15, 3840)
    assert 2 * p["smem"] <= 160 * 1024                        # two workgroups per CU
    covered = 0
    for fi, fo in [(48000, 16000), (32000, 16000), (48000, 24000), (16000, 8000), (44100, 22050)]:
        for ch in (1, 2):
            for K in (1, 2, 16, 17, 33, 64, 65, 66, 100, 128, 129):
                p = plan(fi, fo, ch, K)
                assert p["ok"] == 1, (fi, fo, ch, K)
                covered += 1
                assert p["H2"] % 16 == 0 and p["H2"] >= K - 1 and p["pad"] == p["H2"] - (K - 1) and 0 <= p["pad"] < 16
                assert 4 * p["NKB"] >= K + 15                 # the FIR window of a 16-output block fits the chain
                assert p["NKB"] == (20 if K <= 65 else 36)
                assert p["P2"] == 15 and p["z_per_tile"] == 256 * p["P2"]
                assert p["seg_len_b"] == 16 * p["P2"] + 4 * p["NKB"] and p["seg_stride_b"] % 2 == 1
                # every window sample phase B reads was produced by phase A of the same tile (4096 samples)
                assert 15 * 16 * p["P2"] + p["pad"] + p["seg_len_b"] <= 4096
                # a resampled sample belongs to at most two column segments
                assert p["seg_len_b"] <= 2 * 16 * p["P2"]
    assert covered == 110
    # not covered: long FIRs, ratios with several block matrices or another tile shape, no resampling, no FIR
    assert plan(48000, 16000, 2, 130)["ok"] == 0 and plan(48000, 16000, 1, 1024)["ok"] == 0
    assert plan(48000, 32000, 2, 129)["ok"] == 0 and plan(48000, 44100, 1, 129)["ok"] == 0
    # ten block matrices (44.1 k -> 16 k; resfir10_kernel, round 5): a tile is 5 120 resampled samples AND 5 120 FIR outputs
    # (20 block periods per column), nothing is recomputed (H2 = 0: the K - 1 samples in front of a tile are carried in LDS)
    for ch in (1, 2):
        for K in (1, 17, 65, 66, 129):
            p = plan(44100, 16000, ch, K)
            assert (p["ok"], p["NKA"], p["H2"], p["pad"], p["P2"], p["z_per_tile"]) == (1, 28, 0, 0, 20, 5120), (ch, K)
            assert p["NKB"] == (20 if K <= 65 else 36) and 4 * p["NKB"] >= K + 15
            assert p["seg_len_b"] == 16 * (p["P2"] - 1) + 4 * p["NKB"] and p["seg_stride_b"] % 2 == 1
            assert p["seg_len_b"] - 16 * p["P2"] >= K - 1 and p["smem"] <= 160 * 1024      # a segment's head holds the whole history
    assert plan(44100, 16000, 1, 130)["ok"] == 0
    assert plan(16000, 48000, 1, 129)["ok"] == 0
    assert plan(16000, 16000, 1, 129)["ok"] == 0 and plan(48000, 16000, 2, 0)["ok"] == 0
