"""-m gpu: a short run of the randomised parity sweep (`tests/fuzz_parity.py`): random family / batch /
length / raggedness / speakers / length_scale / max_len / SDP / split-K against the oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(1600)
def test_fuzz_parity_short():
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "fuzz_parity.py"), "64", "7"],
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "fuzz: 64 cases" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
