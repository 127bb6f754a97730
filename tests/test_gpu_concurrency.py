"""Several host threads, each with handles of its own, on one device at the same time (the reference keeps its state per
object too: resampler.rs:12-21, vad.rs:60-70).  What the library shares between handles -- the per-kernel LDS-attribute
flags, the per-device resampler staging, the thread-local error slot, the process-wide switches -- must not leak from one
thread's results into another's: every thread's rows equal the rows the same call gives alone, bit for bit."""
import threading

import numpy as np
import pytest

from util import bits_equal, synth_clip

pytestmark = pytest.mark.gpu


def test_handles_on_four_threads_at_once(af):
    torch = pytest.importorskip("torch")
    n = 100_000
    taps = (np.hamming(129) / np.hamming(129).sum()).astype(np.float32)
    clips = [synth_clip(1000 + t, n, 2, 48000) for t in range(4)]

    def work(t, kernel):
        x = clips[t]
        feat = af.FeaturePipeline(sample_rate=48000, channels=2)
        if kernel:
            assert af.library().af_features_set_option(feat._h, b"kernel", kernel) == 0
        feat.set_batch_uniform(1, n)
        chain = af.ChainPipeline(input_rate=48000, channels=2, output_rate=16000, fir_taps=taps)
        chain.set_batch_uniform(1, n)
        mono = np.ascontiguousarray(x.reshape(-1, 2).mean(axis=1, dtype=np.float32))
        out = []
        for _ in range(6):
            out.append((feat.run(x), chain.run(x), af.resample_batch(mono[None, :], 48000, 44100)[0]))
        return out

    # alone, one after the other
    ref = [work(t, 3 if t == 3 else 0)[0] for t in range(4)]
    got, errs = [None] * 4, []

    def runner(t):
        try:
            got[t] = work(t, 3 if t == 3 else 0)
        except Exception as e:                      # noqa: BLE001 - reported below
            errs.append((t, repr(e)))

    threads = [threading.Thread(target=runner, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errs, errs
    torch.cuda.synchronize()
    for t in range(4):
        for rep in got[t]:
            for a, b in zip(rep, ref[t]):
                assert bits_equal(a, b), t


def test_process_wide_switches_flipped_while_handles_run(af):
    """af_debug_set is process-wide (atomics, read once per call).  Flipped by one thread while three others run:
    handles that set their own option (af_features_set_option / af_chain_set_option) never notice -- their rows are the
    rows of the call alone, bit for bit; a handle that follows the switch gives, call by call, the rows of one of the two
    forms the switch chooses between, never a mixture."""
    torch = pytest.importorskip("torch")
    lib = af.library()
    n = 100_000
    taps = (np.hamming(129) / np.hamming(129).sum()).astype(np.float32)
    x = synth_clip(2000, n, 2, 48000)

    def feat_rows(kernel):                          # kernel: 2 / 1 own option, None: follows the switch
        f = af.FeaturePipeline(sample_rate=48000, channels=2)
        if kernel is not None:
            assert lib.af_features_set_option(f._h, b"kernel", kernel) == 0
        f.set_batch_uniform(1, n)
        return f

    ref2, ref1 = feat_rows(2).run(x), feat_rows(1).run(x)
    assert not bits_equal(ref2, ref1)               # the two kernels differ in the last bits: the test can tell them apart
    chain = af.ChainPipeline(input_rate=48000, channels=2, output_rate=16000, fir_taps=taps)
    chain.set_option("unfused", 0)
    chain.set_option("kernel", 2)                   # its STFT / mel stage is a feature handle: pinned too
    chain.set_batch_uniform(1, n)
    ref_chain = chain.run(x)
    stop = threading.Event()
    errs = []

    def flipper():
        k = 0
        while not stop.is_set():
            k ^= 1
            assert lib.af_debug_set(b"features.kernel", k) == 0
            assert lib.af_debug_set(b"chain.unfused", k) == 0

    def own_option():
        try:
            f = feat_rows(2)
            for _ in range(12):
                assert bits_equal(f.run(x), ref2)
                assert bits_equal(chain.run(x), ref_chain)
        except Exception as e:                      # noqa: BLE001
            errs.append(repr(e))

    def follower():
        try:
            f = feat_rows(None)
            for _ in range(12):
                r = f.run(x)
                assert bits_equal(r, ref2) or bits_equal(r, ref1)
        except Exception as e:                      # noqa: BLE001
            errs.append(repr(e))

    threads = [threading.Thread(target=flipper), threading.Thread(target=own_option), threading.Thread(target=follower)]
    for th in threads:
        th.start()
    for th in threads[1:]:
        th.join()
    stop.set()
    threads[0].join()
    assert lib.af_debug_set(b"features.kernel", 0) == 0 and lib.af_debug_set(b"chain.unfused", 0) == 0
    torch.cuda.synchronize()
    assert not errs, errs


def test_error_text_is_per_thread(af):
    """af_last_error() is thread-local: a failing call on one thread does not replace the other thread's message."""
    lib = af.library()
    seen = {}

    def bad(key, hop):
        try:
            af.FeaturePipeline(sample_rate=48000, channels=2, hop=hop)
        except af.AudioError as e:
            seen[key] = str(e)

    t1 = threading.Thread(target=bad, args=("a", 3))
    t2 = threading.Thread(target=bad, args=("b", 4096))
    t1.start(); t2.start(); t1.join(); t2.join()
    assert "3" in seen["a"] and "4096" in seen["b"]
    assert lib.af_last_error() is not None
