"""CPU: the host code that needs no GPU (fusion, list merge, MockEmbedder, error string) and the C
oracle, built with AddressSanitizer + UndefinedBehaviorSanitizer and driven by
tests/cpp/test_host_sanitize.cpp (`make -C trueno-rag_amd/csrc asan`)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_code_is_clean_under_asan_and_ubsan():
    out = subprocess.run(["make", "-C", os.path.join(ROOT, "trueno-rag_amd", "csrc"), "asan"], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert "all checks passed" in out.stdout
