"""The host side under AddressSanitizer + UndefinedBehaviorSanitizer (the reference's code is
deliberately unchecked; ours is checked on the CPU -- GPU ASan is not available on this pool).

Builds `oracle/libaha_oracle_asan.so` and `aha_amd/libaha_hip_asan.so` (automaton.cpp, capi.cpp and group.cpp with
g++ -fsanitize=address,undefined; the kernel launchers are stubs, every test runs HOST_ONLY) and re-runs the
host-logic, filter-twin and oracle suites against them in a child process with the ASan runtime preloaded.  Any
report aborts the child (halt_on_error) and fails this test."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    p = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(os.environ.get("AHA_HIP_LIB") is not None, reason="already inside the sanitizer run")
def test_host_side_under_asan_ubsan():
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("no libasan in this toolchain")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libaha_oracle_asan.so"])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "aha_amd", "csrc"), "asan"])
    env = dict(os.environ)
    env.update({
        "LD_PRELOAD": asan,
        "ASAN_OPTIONS": "detect_leaks=0:halt_on_error=1:abort_on_error=1",
        "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1",
        "AHA_HIP_LIB": os.path.join(ROOT, "aha_amd", "libaha_hip_asan.so"),
        "AHA_ORACLE_LIB": os.path.join(ROOT, "oracle", "libaha_oracle_asan.so"),
    })
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider",
           os.path.join(ROOT, "tests", "test_host_logic.py"), os.path.join(ROOT, "tests", "test_unit_twin.py"),
           os.path.join(ROOT, "tests", "test_oracle_kats.py"), os.path.join(ROOT, "tests", "test_oracle_vs_model.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail
