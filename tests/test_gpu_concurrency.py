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
