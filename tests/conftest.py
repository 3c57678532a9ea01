import gzip
import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(HERE, "golden")
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN, "kzg_mainnet.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def kats():
    with open(os.path.join(GOLDEN, "kats.json")) as f:
        return json.load(f)


_blob_cache = {}


def load_blob(ref):
    if ref not in _blob_cache:
        with gzip.open(os.path.join(GOLDEN, "blobs", ref + ".bin.gz"), "rb") as f:
            _blob_cache[ref] = f.read()
    return _blob_cache[ref]


@pytest.fixture(scope="session")
def blob_loader():
    return load_blob


@pytest.fixture(scope="session")
def trusted_setup_text():
    with open(os.path.join(GOLDEN, "trusted_setup.txt"), "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def oracle():
    import oracle_ffi

    return oracle_ffi


@pytest.fixture(scope="session")
def oracle_settings(oracle, trusted_setup_text):
    rc, s = oracle.load_settings(trusted_setup_text)
    assert rc == 0
    return s


def load_package():
    """import rust-kzg_amd/ (hyphenated directory) under the module name rust_kzg_amd"""
    import importlib.util

    if "rust_kzg_amd" in sys.modules:
        return sys.modules["rust_kzg_amd"]
    # tests that also use torch for device buffers need torch's HIP runtime initialised before the
    # library pulls in libamdhip64 (both resolve the same SONAME; first loaded wins)
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    path = os.path.join(ROOT, "rust-kzg_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location("rust_kzg_amd", path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["rust_kzg_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def kzg():
    return load_package()
