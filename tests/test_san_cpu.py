"""CPU test: the sanitizer builds of the host code (airlift_amd/csrc/Makefile, target `san`): AddressSanitizer + UBSan and
ThreadSanitizer builds of the threaded host layers run tests/csrc/san_main.cpp -- whole-file parallel FASTA loader, block-parallel
FASTQ parser on regular and irregular files, ordered multi-lane output (offsets + pwrite / turns), the SAM formatter, a corrupt BAM
through the read extraction.  Any report (halt_on_error) or failed self-test fails the build target.  (The reference has no such
build; SURVEY.md 5 lists a live UB in its fork, map.c:315.)"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_host_code_under_asan_ubsan_and_tsan():
    r = subprocess.run(["make", "-s", "-j4", "-C", os.path.join(ROOT, "airlift_amd", "csrc"), "san"], capture_output=True, timeout=1500)
    out = (r.stdout + r.stderr).decode(errors="replace")
    assert r.returncode == 0, out[-4000:]
    assert out.count("san self-tests: 0 failure(s)") == 2, out[-2000:]
    assert "WARNING: ThreadSanitizer" not in out and "ERROR: AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
