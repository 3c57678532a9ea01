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


# Two builds of the library run the suite (rust-kzg_amd/build.py):
#   product  libkzg_mi355x.so        — what ships;
#   exact    libkzg_mi355x_exact.so  — the same sources with -DKZGAMD_FORCE_EXACT_TESTS: the exact zero test that 2.4e-7
#            of the point additions reach (with its LDS exchange and wave-local synchronisation) runs on every one.
# Every test that takes the `kzg` fixture runs once per flavour; KZGAMD_LIB carries the flavour to the processes a test
# starts (fuzzers, ranks, bench.py), and the oracle's answers — not the other flavour's — are what both are held to.
FLAVOURS = {"product": "libkzg_mi355x.so", "exact": "libkzg_mi355x_exact.so"}


def load_package(flavour=None):
    """import rust-kzg_amd/ (hyphenated directory) under the module name rust_kzg_amd (rust_kzg_amd_exact for the
    forced-rare-path build).  flavour None (tools, helper processes): whatever KZGAMD_LIB names, else the product library."""
    import importlib.util

    if flavour is None and os.environ.get("KZGAMD_LIB"):
        flavour = next((k for k, v in FLAVOURS.items() if os.environ["KZGAMD_LIB"].endswith("/" + v)), "custom")
    flavour = flavour or "product"
    name = "rust_kzg_amd" if flavour == "product" else "rust_kzg_amd_" + flavour
    if name in sys.modules:
        return sys.modules[name]
    # tests that also use torch for device buffers need torch's HIP runtime initialised before the
    # library pulls in libamdhip64 (both resolve the same SONAME; first loaded wins)
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    path = os.path.join(ROOT, "rust-kzg_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    saved = os.environ.get("KZGAMD_LIB")
    if flavour != "custom":
        os.environ["KZGAMD_LIB"] = os.path.join(ROOT, "rust-kzg_amd", "csrc", FLAVOURS[flavour])
    try:
        spec.loader.exec_module(mod)
    finally:
        if saved is None:
            del os.environ["KZGAMD_LIB"]
        else:
            os.environ["KZGAMD_LIB"] = saved
    return mod


def _flavours():
    only = os.environ.get("KZGAMD_TEST_FLAVOURS")  # e.g. "product" while iterating on one kernel
    return only.split(",") if only else list(FLAVOURS)


def _ensure_built(flavour):
    """a checkout that has not been through __graft_entry__.build() yet: compile the flavour now (hipcc cross-compiles
    without a GPU); a library that is there is used as it is — its content stamp is build()'s business"""
    import importlib.util

    if os.path.exists(os.path.join(ROOT, "rust-kzg_amd", "csrc", FLAVOURS[flavour])):
        return
    spec = importlib.util.spec_from_file_location("rust_kzg_amd_build", os.path.join(ROOT, "rust-kzg_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    (b.build_exact if flavour == "exact" else b.build)()


@pytest.fixture(scope="session", params=_flavours())
def kzg(request):
    _ensure_built(request.param)
    mod = load_package(request.param)
    saved = os.environ.get("KZGAMD_LIB")
    os.environ["KZGAMD_LIB"] = mod.LIB_PATH
    yield mod
    if saved is None:
        os.environ.pop("KZGAMD_LIB", None)
    else:
        os.environ["KZGAMD_LIB"] = saved
