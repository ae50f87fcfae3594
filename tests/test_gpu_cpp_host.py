"""GPU: the C++ host mirror (trueno-rag_amd/host/trueno_rag.hpp) runs the
reference's unit tests through the C ABI in a process of its own."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_cpp_host_mirror_binary():
    exe = os.path.join(ROOT, "trueno-rag_amd", "lib", "test_host_mirror")
    assert os.path.exists(exe), "build it with make -C trueno-rag_amd/csrc"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all checks passed" in out.stdout
