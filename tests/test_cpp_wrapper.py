"""Builds and runs the C++ twin of spec/ac_spec.cr (tests/cpp/spec_ac.cpp)
against libaha_hip.so through the header-only wrapper include/aha/ac.hpp."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "spec_ac")
    cmd = ["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "spec_ac.cpp"),
           "-L", os.path.join(ROOT, "aha_amd"), "-laha_hip", "-Wl,-rpath," + os.path.join(ROOT, "aha_amd"),
           "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.check_call(cmd)
    return exe


def test_cpp_wrapper_compiles(tmp_path):
    _build(tmp_path)


@pytest.mark.gpu
def test_cpp_spec_passes_on_gpu(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
