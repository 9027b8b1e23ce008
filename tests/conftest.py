import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The C-ABI library is a build product (git-ignored): build it if a fresh checkout lacks it or a
    source is newer (hipcc cross-compiles gfx950 without a GPU; `csrc/build.sh` is incremental)."""
    from mb_istft_vits_amd import _capi
    csrc = _capi.CSRC
    stale = not os.path.isfile(_capi.LIB_PATH)
    if not stale:
        t = os.path.getmtime(_capi.LIB_PATH)
        srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h"))]
        srcs.append(os.path.join(ROOT, "include", "mbistft_vits.h"))
        stale = any(os.path.getmtime(f) > t for f in srcs)
    if stale:
        _capi.build()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
