"""The reference-side patch (integration/denseindex.patch: the DenseIndex trait, defaulted type parameters, the
builder's dense_index()) must apply to the reference as it is. A dry run: nothing is written. Skipped where the
reference or `patch` is absent (the GPU box)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")) or shutil.which("patch") is None,
                    reason="needs the reference tree and patch(1)")
def test_patch_applies_cleanly_to_the_reference():
    r = subprocess.run(["patch", "-p1", "--dry-run", "-d", REF, "-i", os.path.join(ROOT, "integration", "denseindex.patch")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "FAILED" not in r.stdout and "fuzz" not in r.stdout and "offset" not in r.stdout, r.stdout
    for f in ("src/lib.rs", "src/retrieve.rs", "src/pipeline.rs"):
        assert f in r.stdout
