"""GPU parity of the polyphase resampler and FIR (docs/OPS.md, [B]) and of the
reference's resampler bookkeeping (resampler.rs:32-167) through the C ABI.
Values: ||y - ref||_2 / ||ref||_2 <= 1e-5 against float64; lengths, chunk counts
and streaming-vs-one-shot equality are exact."""
import os

import numpy as np
import pytest

from util import bits_equal, rel_err, synth_clip

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-5


@pytest.mark.parametrize("name", ["resample_44k1_16k", "resample_48k_16k", "resample_16k_48k"])
def test_resample_golden(af, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    y = af.resample_batch(g["x"], int(g["in_hz"]), int(g["out_hz"]))[0]
    assert y.size == g["y"].size
    assert rel_err(y, g["y"]) < TOL


@pytest.mark.parametrize("fi,fo", [(44100, 16000), (48000, 16000), (16000, 48000), (48000, 44100),
                                   (8000, 16000), (22050, 16000),
                                   # L = 15: column segments shorter than the workgroup (960 threads)
                                   (48000, 45000), (8000, 15000)])
@pytest.mark.parametrize("n", [1, 31, 32, 33, 1000, 20011])
def test_resample_batch_vs_oracle(af, po, fi, fo, n):
    clips = np.stack([synth_clip(50 + i, n, 1, fi) for i in range(3)])
    y = af.resample_batch(clips, fi, fo)
    L, M = po.resample_ratio(fi, fo)
    assert y.shape == (3, po.resample_out_len(n, L, M))          # ceil(n*L/M), exact
    for i in range(3):
        ref = po.resample(clips[i], fi, fo)
        assert np.linalg.norm(y[i] - ref) <= TOL * max(np.linalg.norm(ref), 1e-30) + 1e-12


@pytest.mark.parametrize("fi,fo", [(48000, 44100), (44100, 48000), (48000, 22050), (16000, 44100), (96000, 44100),
                                   (44100, 32000)])
@pytest.mark.parametrize("n", [2559, 81920 + 5, 200003])
def test_resample_many_block_matrices_vs_oracle(af, po, fi, fo, n):
    """Ratios whose L has no factor in common with 16 need L (147, 441, 320 ...) block matrices: the streamed
    banded kernel (tiles = a range of matrices x 32 block periods; 48 -> 44.1 k: 2560 input samples per period,
    81 920 per tile).  Lengths below one period, just past one tile and over several tiles."""
    import ctypes as C
    info = (C.c_int32 * 12)()
    L, M = po.resample_ratio(fi, fo)
    clips = np.stack([synth_clip(150 + i, n, 1, fi) for i in range(2)])
    y = af.resample_batch(clips, fi, fo)
    assert y.shape == (2, po.resample_out_len(n, L, M))
    for i in range(2):
        ref = po.resample(clips[i], fi, fo)
        assert np.linalg.norm(y[i] - ref) <= TOL * max(np.linalg.norm(ref), 1e-30) + 1e-12
    # and against the thread-per-output kernel it replaces (different summation order: tolerance, not bits)
    lib = af.library()
    try:
        assert lib.af_debug_set(b"banded.off", 1) == 0
        y2 = af.resample_batch(clips, fi, fo)
    finally:
        lib.af_debug_set(b"banded.off", 0)
    assert np.linalg.norm(y - y2) <= 2e-6 * np.linalg.norm(y2)


@pytest.mark.parametrize("fi,fo", [(48000, 16000), (44100, 16000), (16000, 48000), (48000, 44100)])
@pytest.mark.parametrize("n", [1, 2, 5, 1000, 30011])
def test_resample_cubic_mode_vs_oracle(af, po, fi, fo, n):
    """The optional cubic mode (what resampler.rs:43-49 configures; docs/OPS.md, parity unpinned): output length as the
    polyphase resampler, values within 1e-6 of the peak of the float64 formula; a slow sine is reproduced closely."""
    clips = np.stack([synth_clip(170 + i, n, 1, fi) for i in range(2)])
    y = af.resample_cubic_batch(clips, fi, fo)
    L, M = po.resample_ratio(fi, fo)
    assert y.shape == (2, po.resample_out_len(n, L, M))
    for i in range(2):
        ref = po.resample_cubic(clips[i], fi, fo)
        assert np.max(np.abs(y[i] - ref)) <= 1e-6 * max(np.max(np.abs(ref)), 1e-30) + 1e-9
    if n >= 1000:
        t = np.arange(n) / fi
        x = (0.5 * np.sin(2 * np.pi * 220.0 * t)).astype(np.float32)
        yy = af.resample_cubic_batch(x, fi, fo)[0]
        want = 0.5 * np.sin(2 * np.pi * 220.0 * np.arange(yy.size) / fo)
        assert np.max(np.abs(yy[10:-20] - want[10:-20])) < 2e-4   # interpolation error of a 220 Hz tone (clip edges: zero extension)


def test_resample_reference_surface(af):
    r = af.AudioResampler.new(48000, 16000)                      # resampler.rs:192-198
    assert r.input_rate() == 48000 and r.output_rate() == 16000 and r.needs_resampling()
    assert not af.AudioResampler.new(48000, 48000).needs_resampling()   # :200-204
    p = af.AudioResampler.new(16000, 16000)                      # :184-190 bit-identical passthrough
    x = np.array([0.1, 0.2, 0.3, 0.4], np.float32)
    assert bits_equal(p.process(x), x)
    d = af.AudioResampler.default()                              # :169-178
    assert (d.input_rate(), d.output_rate()) == (48000, 16000)
    assert af.AudioResampler.create_48k_to_16k().needs_resampling()   # :60-62


@pytest.mark.parametrize("fi,fo", [(48000, 16000), (44100, 16000), (16000, 48000), (48000, 44100), (16000, 44100),
                                   (48000, 22050), (44100, 48000)])
def test_streaming_equals_one_shot_bit_exact(af, po, fi, fo):
    # (chunks start at any output index: the kernels' 16-byte store paths -- deferred stores of the streamed kernel,
    # buffer stores of the ten-wave shapes -- must hand unaligned chunks to the predicated stores)
    x = synth_clip(60, 10007 if fo not in (44100, 22050) else 100003, 1, fi)
    one = af.resample_batch(x, fi, fo)[0]
    r = af.AudioResampler.new(fi, fo)
    rng = np.random.default_rng(1)
    parts, pos = [], 0
    while pos < x.size:
        n = int(rng.integers(0, 700))
        parts.append(r.process(x[pos:pos + n]))
        pos += n
    parts.append(r.flush())
    got = np.concatenate(parts)
    assert got.size == one.size                                   # ceil(n*L/M)
    assert bits_equal(got, one)                               # same arithmetic per output index
    assert rel_err(got, po.resample(x, fi, fo)) < TOL


def test_streaming_latency_bookkeeping(af, po):
    # an output is released once input sample floor(m*M/L) + T/2 has arrived
    r = af.AudioResampler.new(48000, 16000)
    x = synth_clip(61, 128 * 5, 1, 48000)
    total = 0
    for c in range(5):
        total += r.process(x[c * 128:(c + 1) * 128]).size
        n_seen = 128 * (c + 1)
        assert total == -(-(n_seen - 32) // 3)                    # ceil((n - T/2) * L / M)
    assert total + r.flush().size == po.resample_out_len(640, 1, 3)


def test_batch_resampler_chunking(af, po):
    # resampler.rs:132-166: whole 128-frame chunks, carry, zero-padded flush
    b = af.BatchResampler.new(48000, 16000)
    inner = af.AudioResampler.new(48000, 16000)
    x = synth_clip(62, 1000, 1, 48000)
    outs, fed, carry = [], 0, 0
    for n in (100, 28, 300, 1, 127, 444):
        seg = x[fed:fed + n]
        outs.append(b.process(seg))
        n_chunks, carry = po.chunker_process(carry, n, 128)
        assert b.buffered() == carry                              # exact carry bookkeeping
        fed += n
    assert fed == 1000 and carry == 1000 % 128
    outs.append(b.flush())
    assert b.buffered() == 0
    assert b.flush().size == 0                                    # :153 empty buffer -> nothing
    got = np.concatenate(outs)
    # the inner resampler saw 8 chunks: 7 full + the remainder zero-padded to 128
    padded = np.concatenate([x, np.zeros(8 * 128 - 1000, np.float32)])
    want = inner.process(padded)
    assert bits_equal(got, want)
    assert got.size == -(-(1024 - 32) // 3)


def test_batch_resampler_equal_rates_is_passthrough(af):
    # the reference never terminates here (chunk_size 0); documented deviation: passthrough
    b = af.BatchResampler.new(16000, 16000)
    x = np.arange(10, dtype=np.float32)
    assert bits_equal(b.process(x), x)
    assert b.flush().size == 0


def test_resample_config3_full_size(af, po):
    # BASELINE config 3 in full: 4096 mono clips x 441 000 samples @44.1 k (7.2 GB) -> 160 000 @16 k:
    # every output written, duplicate clips identical, determinism, sampled clips against the oracle
    torch = pytest.importorskip("torch")
    import ctypes as C
    dev = torch.device("cuda:0")
    n_clips, n_in = 4096, 441000
    lib = af.library()
    n_out = lib.af_resample_out_len(44100, 16000, n_in)
    assert n_out == 160000
    gen = torch.Generator(device=dev)
    gen.manual_seed(77)
    x = torch.rand(n_clips, n_in, device=dev, generator=gen).sub_(0.5)
    x[4095] = x[17]
    y = torch.full((n_clips, n_out), float("nan"), device=dev)
    y2 = torch.empty_like(y)
    for dst in (y, y2):
        assert lib.af_resample_batch_device(44100, 16000, C.c_void_p(x.data_ptr()), n_clips, n_in,
                                            C.c_void_p(dst.data_ptr()), None) == 0
    torch.cuda.synchronize()
    assert bool(torch.isfinite(y).all()) and bits_equal(y, y2) and bits_equal(y[4095], y[17])
    for c in (0, 2048, 4094):
        ref = po.resample(x[c].cpu().numpy(), 44100, 16000)
        assert rel_err(y[c].cpu().numpy(), ref) < TOL


def test_resample_48k_to_44k1_full_size(af, po):
    # the everyday device pair (capture.rs:60-80 defaults to 48 kHz; config 3 is 44.1 kHz) at batch size: 1024 mono
    # clips x 30 s @48 k (5.9 GB) -> 1 323 000 samples @44.1 k through the streamed banded kernel (147 block
    # matrices): every output written, duplicate clips identical, determinism, sampled clips against the oracle,
    # DC gain 1 in the interior
    torch = pytest.importorskip("torch")
    import ctypes as C
    dev = torch.device("cuda:0")
    n_clips, n_in = 1024, 1440000
    lib = af.library()
    n_out = lib.af_resample_out_len(48000, 44100, n_in)
    assert n_out == 1323000
    gen = torch.Generator(device=dev)
    gen.manual_seed(78)
    x = torch.rand(n_clips, n_in, device=dev, generator=gen).sub_(0.5)
    x[1023] = x[9]
    x[500] = 0.25
    y = torch.full((n_clips, n_out), float("nan"), device=dev)
    y2 = torch.empty_like(y)
    for dst in (y, y2):
        assert lib.af_resample_batch_device(48000, 44100, C.c_void_p(x.data_ptr()), n_clips, n_in,
                                            C.c_void_p(dst.data_ptr()), None) == 0
    torch.cuda.synchronize()
    assert bool(torch.isfinite(y).all()) and bits_equal(y, y2) and bits_equal(y[1023], y[9])
    assert float((y[500, 100:-100] - 0.25).abs().max()) < 1e-6
    for c in (0, 511, 1022):
        ref = po.resample(x[c].cpu().numpy(), 48000, 44100)
        assert rel_err(y[c].cpu().numpy(), ref) < TOL


def test_resample_full_size_property(af):
    # BASELINE config 3 shapes (a 64-clip slice): DC gain 1 and linearity at full length
    torch = pytest.importorskip("torch")
    import ctypes as C
    dev = torch.device("cuda:0")
    n_clips, n_in = 64, 441000
    lib = af.library()
    n_out = lib.af_resample_out_len(44100, 16000, n_in)
    assert n_out == 160000
    a = torch.rand(n_clips, n_in, device=dev) - 0.5
    b = torch.rand(n_clips, n_in, device=dev) - 0.5
    ya, yb, yab = (torch.empty(n_clips, n_out, device=dev) for _ in range(3))

    def run(x, y):
        assert lib.af_resample_batch_device(44100, 16000, C.c_void_p(x.data_ptr()), n_clips, n_in,
                                            C.c_void_p(y.data_ptr()), None) == 0
    run(a, ya)
    run(b, yb)
    run(a + b, yab)
    one = torch.ones(n_clips, n_in, device=dev)
    y1 = torch.empty(n_clips, n_out, device=dev)
    run(one, y1)
    torch.cuda.synchronize()
    assert float((yab - (ya + yb)).abs().max()) < 2e-6            # linearity
    assert float((y1[:, 64:-64] - 1.0).abs().max()) < 1e-6        # unit DC gain per phase
    assert bool(torch.isfinite(yab).all())


# ---- FIR -----------------------------------------------------------------------------
def test_fir_golden(af):
    g = np.load(os.path.join(GOLD, "fir_129.npz"))
    y = af.fir_batch(g["x"], g["taps"])[0]
    assert rel_err(y, g["y"]) < TOL


@pytest.mark.parametrize("K", [1, 2, 3, 4, 5, 64, 129, 1000, 1024])
@pytest.mark.parametrize("n", [1, 7, 1279, 1280, 1281, 5000])
def test_fir_batch_vs_oracle(af, po, K, n):
    rng = np.random.default_rng(K * 7 + n)
    taps = (rng.standard_normal(K) / np.sqrt(K)).astype(np.float32)
    clips = np.stack([synth_clip(70 + i, n, 1, 16000) for i in range(2)])
    y = af.fir_batch(clips, taps)
    assert y.shape == clips.shape                                 # "same length, causal"
    for i in range(2):
        ref = po.fir(clips[i], taps)
        assert np.linalg.norm(y[i] - ref) <= TOL * max(np.linalg.norm(ref), 1e-30)


def test_fir_streaming_equals_one_shot(af, po):
    taps = np.hamming(129).astype(np.float32) / 70.0
    x = synth_clip(80, 9000, 1, 16000)
    one = af.fir_batch(x, taps)[0]
    f = af.FirFilter.new(taps)
    rng = np.random.default_rng(2)
    parts, pos = [], 0
    while pos < x.size:
        n = int(rng.integers(1, 900))
        parts.append(f.process(x[pos:pos + n]))
        pos += n
    got = np.concatenate(parts)
    assert got.size == x.size
    assert rel_err(got, po.fir(x, taps)) < TOL
    assert np.max(np.abs(got - one)) <= 1e-6 * np.max(np.abs(one))
    f.reset()
    assert rel_err(f.process(x[:500]), po.fir(x[:500], taps)) < TOL


def test_fir_rejects_bad_tap_counts(af):
    with pytest.raises(af.AudioError):
        af.FirFilter.new(np.zeros(0, np.float32))
    with pytest.raises(af.AudioError):
        af.FirFilter.new(np.zeros(1025, np.float32))


# ---- banded-MFMA path specifics (audio-flow-rs_amd/csrc/banded_mfma.hip) -----------------------
@pytest.mark.parametrize("K", [40, 161, 177, 178, 200, 352, 353, 529, 1024])
def test_fir_mfma_boundary(af, po, K):
    # <= 177 taps are one MFMA pass (NK <= 48); longer filters are cut into passes of 176 taps that
    # accumulate into y (banded.hpp FirParts); "banded.off" = the thread-per-output kernel, same results
    rng = np.random.default_rng(K)
    taps = (rng.standard_normal(K) / K).astype(np.float32)
    x = synth_clip(K, 30011, 1, 16000)
    y = af.fir_batch(x.reshape(1, -1), taps)[0]
    ref = po.fir(x, taps)
    assert rel_err(y, ref) < 1e-5
    lib = af.library()
    try:
        assert lib.af_debug_set(b"banded.off", 1) == 0
        assert rel_err(af.fir_batch(x.reshape(1, -1), taps)[0], ref) < 1e-5
    finally:
        lib.af_debug_set(b"banded.off", 0)
    # streaming through the handle: chunking does not change a bit of the multi-pass result
    f = af.FirFilter.new(taps)
    got = np.concatenate([f.process(x[a:b]) for a, b in [(0, 5000), (5000, 5001), (5001, 20000), (20000, x.size)]])
    assert rel_err(got, ref) < 1e-5 and np.max(np.abs(got - y)) <= 2e-6 * np.max(np.abs(y))


@pytest.mark.parametrize("fi,fo,n", [(44100, 16000, 44100 * 3), (44100, 48000, 44100 * 2), (48000, 22050, 96000 * 2),
                                     (48000, 22050, 96000)])
def test_resample_aligned_rows_take_the_16_byte_store_paths(af, po, fi, fo, n):
    """Row pitches that are multiples of four floats: the ten-wave shapes store through the range-checked buffer
    resource, the streamed kernel with ranges of <= 16 matrices through its LDS image of the tile -- same bits as the
    predicated stores a stream's unaligned chunks take, partial tiles at the clip ends included."""
    L, M = po.resample_ratio(fi, fo)
    assert po.resample_out_len(n, L, M) % 4 == 0
    clips = np.stack([synth_clip(170 + i, n, 1, fi) for i in range(3)])
    y = af.resample_batch(clips, fi, fo)
    rng = np.random.default_rng(4)
    for i in range(3):
        r = af.AudioResampler.new(fi, fo)
        parts, pos = [], 0
        while pos < n:
            k = int(rng.integers(1, 9000))
            parts.append(r.process(clips[i][pos:pos + k]))
            pos += k
        parts.append(r.flush())
        assert bits_equal(np.concatenate(parts), y[i])
        assert rel_err(y[i], po.resample(clips[i], fi, fo)) < TOL


@pytest.mark.parametrize("fi,fo", [(44100, 16000), (48000, 16000), (44100, 48000), (8000, 16000)])
def test_resample_many_ragged_tiles(af, po, fi, fo):
    # several clips whose lengths leave partial tiles, partial blocks and an unaligned row pitch
    n = 50021
    clips = np.stack([synth_clip(100 + i, n, 1, fi) for i in range(5)])
    out = af.resample_batch(clips, fi, fo)
    for i in range(5):
        ref = po.resample(clips[i], fi, fo)
        assert out[i].size == ref.size
        assert rel_err(out[i], ref) < 1e-5


def test_fir_streaming_chunk_invariance_mfma(af, po):
    # the MFMA tile grid is anchored at absolute sample 0: any chunking gives the same bits
    rng = np.random.default_rng(7)
    taps = (rng.standard_normal(129) / 129).astype(np.float32)
    x = synth_clip(77, 40000, 1, 16000)
    whole = af.FirFilter.new(taps).process(x)
    f = af.FirFilter.new(taps)
    parts, pos = [], 0
    for step in [1, 15, 16, 17, 4095, 4096, 4097, 9000, 40000]:
        parts.append(f.process(x[pos:pos + step]))
        pos += step
        if pos >= x.size:
            break
    got = np.concatenate(parts)
    assert bits_equal(got, whole[:got.size])
    assert rel_err(whole, po.fir(x, taps)) < 1e-5


@pytest.mark.parametrize("fi,fo", [(48000, 16000), (16000, 48000), (8000, 16000), (16000, 16000), (44100, 16000),
                                   (44100, 48000)])
@pytest.mark.parametrize("n", [1, 33, 1000, 30011])
def test_fused_stereo_downmix_resample(af, po, fi, fo, n):
    # to_mono (capture.rs:30-42) inside the resampler's loads == to_mono, then resample
    clips = np.stack([synth_clip(300 + i, n, 2, fi) for i in range(3)])
    assert af.library().af_resample_has_stereo_path(fi, fo) == 1
    y = af.resample_stereo_batch(clips, fi, fo)
    for i in range(3):
        mono = po.to_mono(clips[i], 2)
        ref = po.resample(mono, fi, fo) if fi != fo else mono
        assert y[i].size == ref.size
        assert np.linalg.norm(y[i] - ref) <= TOL * max(np.linalg.norm(ref), 1e-30) + 1e-12


def test_fused_stereo_path_is_reported_per_ratio(af):
    lib = af.library()
    assert lib.af_resample_has_stereo_path(44100, 16000) == 1     # ten-wave shape: one workgroup per CU, still one pass
    assert lib.af_resample_has_stereo_path(48000, 44100) == 0     # streamed matrices: separate to_mono


def test_resample_batch_host_path_pipelined_from_pinned_memory(af):
    """af_resample_batch from page-locked memory cuts the batch into clip chunks (H2D / kernel / D2H on three
    streams); from pageable memory it is one copy each way.  Same bits either way, tail chunk included."""
    torch = pytest.importorskip("torch")
    import ctypes as C
    lib = af.library()
    n_clips, n_in = 45, 480000                                     # 86 MB in: 3 chunks of 17 + 11 clips
    n_out = lib.af_resample_out_len(48000, 16000, n_in)
    g = torch.Generator()
    g.manual_seed(5)
    x = torch.rand(n_clips, n_in, generator=g) - 0.5
    ref = af.resample_batch(x.numpy(), 48000, 16000)               # pageable: the plain path
    xp = x.pin_memory()
    out = torch.empty(n_clips, n_out).pin_memory()
    st = lib.af_resample_batch(48000, 16000, C.c_void_p(xp.data_ptr()), n_clips, n_in, C.c_void_p(out.data_ptr()),
                               out.numel())
    assert st == 0
    assert bits_equal(out.numpy(), ref)
