"""CPU checks of the hop-256 kernel's FFT structure (no GPU):
  * audio-flow-rs_amd/csrc/fft_regs.hpp -- the FMA-shaped register FFTs, compiled on the host with g++
    (tests/cpp/fft_regs_test.cpp) against a float64 DFT;
  * the 16 x 16 x 4 index maps of features_v2.hip (stage 1 / T1 / stage 2a / T2 / stage 2b, the partner
    lanes of the packed-spectrum split), replayed in numpy (scripts/proto/fft_r16x16x4.py)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_register_ffts_on_the_host(tmp_path):
    exe = str(tmp_path / "fft_regs_test")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", "fft_regs_test.cpp")],
                   check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    errs = [float(line.split()[-1]) for line in r.stdout.splitlines() if line.startswith("N=")]
    assert len(errs) == 4 and max(errs) < 5e-7                      # N = 4, 8, 16, 32


def test_v2_index_maps_in_numpy():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "proto", "fft_r16x16x4.py")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    fft_err = float(r.stdout.splitlines()[0].split()[-1])
    split = [float(x) for x in r.stdout.splitlines()[1].split()[-2:]]
    assert fft_err < 1e-10 and max(split) < 1e-9
