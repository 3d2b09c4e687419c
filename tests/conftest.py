"""pytest configuration: markers, paths, shared fixtures."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The product has no fallback, so even the CPU tests (symbol export, host packer) need libmi_nerf.so: build it when the
    tree is fresh (hipcc cross-compiles gfx950 without a GPU; a no-op when the library is up to date)."""
    from nerf_pytorch_paeng_amd.build import build_library
    return build_library()


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def fake_rccl_lib(tmp_path_factory):
    """tests/c_abi/fake_rccl.cpp built once per session: a stand-in for librccl (shared-memory exchange between ranks that SHARE one GPU -- RCCL itself
    refuses two ranks on a device), named to the library by MI_NERF_RCCL_LIB."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    lib = str(tmp_path_factory.mktemp("fake_rccl") / "libfake_rccl.so")
    r = subprocess.run([hipcc, "-shared", "-fPIC", "-O1", "-x", "hip", "--offload-arch=gfx950", os.path.join(ROOT, "tests", "c_abi", "fake_rccl.cpp"), "-o", lib, "-lrt"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return lib


@pytest.fixture(scope="session")
def oracle_cache():
    """Results of the CPU oracle that more than one GPU test needs for the same inputs (BASELINE config #2, all 4096 rays, ~6 s of CPU per
    evaluation): computed by whoever asks first, keyed by the caller."""
    return {}
