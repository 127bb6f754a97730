"""The C++ host mirror (include/audioflow.hpp) exercised by a C++ twin of the reference's unit
tests (tests/cpp/harness.cpp): compiled with g++ against the in-tree C-ABI library and run on
the GPU box.  This is the same boundary the Rust shim in INTEGRATION.md binds."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp):
    exe = os.path.join(tmp, "harness")
    libdir = os.path.join(ROOT, "audio-flow-rs_amd")
    cmd = ["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "harness.cpp"), "-o", exe,
           "-L", libdir, "-laudioflow_dsp", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True)
    return exe


def test_cpp_mirror_compiles_against_the_c_abi(tmp_path):
    # no GPU needed: the header is self-contained C++17 over include/audioflow_dsp.h
    assert os.path.exists(_build(str(tmp_path)))


@pytest.mark.gpu
def test_cpp_harness_passes(tmp_path):
    exe = _build(str(tmp_path))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "harness ok" in r.stdout
