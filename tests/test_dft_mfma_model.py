"""Host-side check of the matrix formulation behind features_v3.hip (scripts/proto/dft_mfma_model.py): the two-stage
32 x 32 DFT with the Hann window and every twiddle folded into the stage-1 matrices, in the kernel's row order, scales
and bin mapping, is the 1024-point rfft exactly; with f16-pair operands (three products per MAC, f32 accumulation) it
stays inside the contract's 1e-5 per frame with two decades of margin, also across a 120 dB level step inside one
16-frame unit.  (The kernel itself is compared with the float64 oracle on the GPU: tests/test_gpu_features_v3.py.)"""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("dft_mfma_model", os.path.join(ROOT, "scripts", "proto", "dft_mfma_model.py"))
model = importlib.util.module_from_spec(spec)
spec.loader.exec_module(model)

L = 15 * model.HOP + model.N


def _err(got, ref):
    return np.linalg.norm(got - ref, axis=1) / np.linalg.norm(ref, axis=1)


def test_two_stage_dft_is_the_rfft():
    rng = np.random.default_rng(0)
    for x in (rng.standard_normal(L) * 0.3, np.sin(2 * np.pi * 440.0 / 48000 * np.arange(L)), np.full(L, 0.9)):
        assert _err(model.power_rows(x, 16, exact=True), model.reference(x, 16)).max() < 1e-12
    # n_fft 512: every second bin of the 1024-point transform of the zero-padded frame
    x = rng.standard_normal(L)
    ref = model.reference(x, 16, 512)
    got = model.power_rows(x, 16, True, 512)[:, ::2]
    assert np.max(np.abs(got - ref)) / np.max(ref) < 1e-12


def test_row_order_covers_every_bin_once():
    bins = []
    for k1 in range(1, 16):
        for r in range(0, 64, 2):
            k2 = model.k2_of(*divmod(r, 16))
            bins.append(k1 + 32 * k2 if k2 < 16 else (32 - k1) + 32 * (31 - k2))
    bins += [32 * k2 for k2 in range(17)] + [16 + 32 * k2 for k2 in range(16)]
    assert sorted(bins) == list(range(513))
    assert sorted(model.k2_of(rb, m) for rb in range(4) for m in range(0, 16, 2)) == list(range(32))


def test_f16_pair_operands_meet_the_tolerance():
    rng = np.random.default_rng(1)
    cases = {
        "noise": rng.standard_normal(L) * 0.3,
        "tone": np.sin(2 * np.pi * 440.0 / 48000 * np.arange(L)) + 1e-3 * rng.standard_normal(L),
        "80 dB step": np.concatenate([rng.standard_normal(L // 2), 1e-4 * rng.standard_normal(L - L // 2)]),
        "120 dB step": np.concatenate([rng.standard_normal(L // 2), 1e-6 * rng.standard_normal(L - L // 2)]),
        "x 1e-12": 1e-12 * rng.standard_normal(L),
    }
    for name, x in cases.items():
        e = _err(model.power_rows(x, 16, exact=False), model.reference(x, 16))
        assert e.max() < 1e-6, (name, e.max())
