import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_package():
    """The package directory is `trueno-rag_amd` (hyphen): import it as trueno_rag_amd."""
    if "trueno_rag_amd" in sys.modules:
        return sys.modules["trueno_rag_amd"]
    pkg_dir = os.path.join(ROOT, "trueno-rag_amd")
    spec = importlib.util.spec_from_file_location(
        "trueno_rag_amd", os.path.join(pkg_dir, "__init__.py"), submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["trueno_rag_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def trag():
    return load_package()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.lib()
    return o


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
