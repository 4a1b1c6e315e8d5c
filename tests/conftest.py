import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    O.lib()
    return O


@pytest.fixture(scope="session")
def awfm():
    """the product library through its C ABI; built in-tree if missing (never falls back to anything else)"""
    from avxwindowfmindex_amd import _lib, api
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    _lib.lib()
    return api


@pytest.fixture(scope="session")
def require_gpu(awfm):
    from avxwindowfmindex_amd import _lib
    n = _lib.lib().awfmGpuDeviceCount()
    assert n > 0, "GPU test selected but no HIP device is visible: the HIP path must run, there is no fallback"
    return n


def set_diag(monkeypatch, **keys):
    """adds key=value pairs to $AWFM_GPU_DIAG (the library's one variable of test and diagnostics hooks: include/awfm_gpu.h);
    a value of None takes the key out"""
    have = dict(kv.split("=", 1) for kv in os.environ.get("AWFM_GPU_DIAG", "").split(",") if "=" in kv)
    for key, value in keys.items():
        if value is None:
            have.pop(key, None)
        else:
            have[key] = str(value)
    if have:
        monkeypatch.setenv("AWFM_GPU_DIAG", ",".join(f"{k}={v}" for k, v in have.items()))
    else:
        monkeypatch.delenv("AWFM_GPU_DIAG", raising=False)


@pytest.fixture
def diag(monkeypatch):
    """diag(key=value, ...): see set_diag"""
    return lambda **keys: set_diag(monkeypatch, **keys)


@pytest.fixture(params=["narrow", "wide", "wide-superblocks"])
def wide(request, monkeypatch):
    """runs a GPU test three times: with 32-bit BWT positions in the kernels (what an index below 2^32 positions
    gets); with the 64-bit instantiations forced on the same small index ($AWFM_GPU_FORCE_WIDE is read when a device
    image is created); and with nucleotide superblocks of a few thousand positions instead of 2^32 on top of that
    ($AWFM_GPU_DIAG nuc_super_shift=auto: up to 48 superblocks), i.e. the code and the base-count arithmetic an index of
    2^32 or more positions runs (ref src/AwFmIndex.h:88-91 is 64-bit throughout)"""
    monkeypatch.setenv("AWFM_GPU_FORCE_WIDE", "0" if request.param == "narrow" else "1")
    if request.param == "wide-superblocks":
        set_diag(monkeypatch, nuc_super_shift="auto")
    return request.param != "narrow"


