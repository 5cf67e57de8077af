"""SURVEY.md section 5 (sanitizers on the CPU side): the C oracle rebuilt with AddressSanitizer + UBSan (`make -C oracle asan`)
and held to the reference's golden fixtures again, in a child interpreter that has libasan preloaded.  Any out-of-bounds
access, use of uninitialised stack, signed overflow or misaligned access in the restatement aborts the child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    try:
        path = subprocess.check_output(["gcc", "-print-file-name=" + name], text=True).strip()
    except (OSError, subprocess.CalledProcessError):
        return None
    return path if os.path.isabs(path) and os.path.exists(path) else None


def test_oracle_under_asan_and_ubsan_passes_the_golden_fixtures():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("gcc sanitizer runtimes not installed")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    env = dict(os.environ)
    env.update(MAD_ORACLE_SANITIZE="1", LD_PRELOAD=asan + ":" + ubsan, PYTHONDONTWRITEBYTECODE="1",
               # python itself leaks by design at exit and numpy's allocators are not instrumented: leak checking off,
               # everything else (heap / stack / global overflows, use after free, UB) aborts
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_golden.py"), "-x", "-q", "-p", "no:cacheprovider"],
                         cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    tail = "\n".join(res.stdout.splitlines()[-25:])
    assert res.returncode == 0, tail
    assert "passed" in tail and "AddressSanitizer" not in res.stdout and "runtime error" not in res.stdout, tail
