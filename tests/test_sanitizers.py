"""AddressSanitizer + UBSan over the CPU-compilable parts of the product (GPU sanitizers are not available
on the pool, so the kernels' scalar numerics header and the whole host layer are exercised on the host)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "harness", "sanitize_driver.cpp")
EXE = os.path.join(ROOT, "tests", "harness", "sanitize_driver")


def test_host_layer_and_core_header_are_clean_under_asan_ubsan():
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=all", "-I", os.path.join(ROOT, "include"), "-o", EXE, SRC])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([EXE], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "sanitize_driver: ok" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
