"""Parity on an index that needs the wide path (>= 2^32 rows; no KAIJU_GPU_FORCE_WIDE).  Building such an index takes
minutes and ~60 GB of host memory, so the test runs only where tests/tools/wide_index.py prepared one:
    python tests/tools/wide_index.py prepare /tmp/kjwide && KAIJU_TEST_WIDE_DIR=/tmp/kjwide pytest -m gpu tests/test_gpu_wide.py
The forced-wide tests on the golden index (test_gpu_parity.py::test_wide_index_path) run everywhere."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_wide_index_parity(gpu_lib):
    W = os.environ.get("KAIJU_TEST_WIDE_DIR")
    if not W or not os.path.exists(os.path.join(W, "db.fmi")):
        pytest.skip("no prepared >= 2^32-row index (KAIJU_TEST_WIDE_DIR, tests/tools/wide_index.py prepare)")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    import wide_index
    res = wide_index.parity(W, sample=20000)
    for name, r in res.items():
        assert r["mismatches"] == 0 and r["error_flags"] == 0, (name, r)
        assert r["with_hit"] > 0.5 * r["checked"], (name, r)
