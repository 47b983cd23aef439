import os
import sys

import pytest

os.environ.setdefault("SA_HIP_DIAG", "1")   # the SA_HIP_* plan switches the tests flip are only read with this set (csrc/common.hpp: diag_env)

try:   # torch first: a process must end up with ONE HIP runtime -- the library that is loaded first decides which
    import torch  # noqa: F401  (tests that hand torch device buffers to the C ABI run in this process)
except Exception:   # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    from oracle.oracle import Ref
    if not Ref.available():
        pytest.skip("oracle/_ref/libsa_ref.so not built (reference sources absent)")
    return Ref()


@pytest.fixture(scope="session")
def capi():
    """The C ABI; building is part of the contract (hipcc cross-compiles without a GPU)."""
    from suffixarray_amd.build import build_lib, build_cython
    build_lib()
    build_cython()
    from suffixarray_amd import _capi
    _capi.lib()
    return _capi


@pytest.fixture(scope="session")
def gpu(capi):
    n = capi.lib().sa_hip_device_count()
    if n < 1:
        pytest.fail("gpu test selected but no HIP device is usable: " + capi.lib().sa_hip_last_error().decode())
    return capi
