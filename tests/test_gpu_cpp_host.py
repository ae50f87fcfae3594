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


@pytest.mark.gpu
@pytest.mark.parametrize("rows,dim", [(10_000, 384), (1000, 128)])
def test_concurrent_searches_on_one_handle_scale(rows, dim):
    """search(&self) is concurrent in the reference (src/index.rs:386). Four threads on ONE handle of the
    smallest benchmark configuration's shape (10 000 x 384, 25 us of GPU work per query: latency-bound)
    and of the reference bench's own (1000 x 128, benches/retrieval.rs:71-94): every result equals the
    one-thread result bit for bit, and the four threads together answer at least twice as many queries
    per second as one (a thread that finds the handle busy searches in a spare context beside it)."""
    import json
    exe = os.path.join(ROOT, "trueno-rag_amd", "lib", "test_concurrent")
    assert os.path.exists(exe), "build it with make -C trueno-rag_amd/csrc"
    out = subprocess.run([exe, str(rows), str(dim), "3000", "4"], capture_output=True, text=True, timeout=180)
    assert out.returncode == 0, out.stdout + out.stderr
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["mismatches"] == 0
    assert r["speedup"] >= 2.0, r
