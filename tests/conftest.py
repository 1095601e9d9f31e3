import os
import subprocess
import sys

import pytest

collect_ignore_glob = ["tools/*", "emu/*"]   # helper scripts, not tests

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import pyoracle as po
    po.build_oracle()
    return po.Oracle()


@pytest.fixture(scope="session")
def emu():
    """the kernel logic of kaiju_amd/csrc/kj_core.h compiled for the host (test infrastructure)"""
    import util
    # KAIJU_EMU_DEFINES="KJ_LOC_ILP=2 ...": the whole emulator suite on a compile-time variant of the lanes (a library of its own)
    defs = tuple(os.environ.get("KAIJU_EMU_DEFINES", "").split())
    if defs:
        tag = "".join(c if c.isalnum() else "_" for c in "_".join(defs))
        return util.Emu(so=os.path.join(util.EMU_DIR, f"libkaiju_kernel_emu_{tag}.so"), defines=defs)
    return util.Emu()


@pytest.fixture(scope="session")
def golden():
    import util
    return util.Golden()


@pytest.fixture(scope="session")
def gpu_lib():
    from kaiju_amd import api
    api.lib()
    if api.device_count() < 1:
        pytest.fail("GPU test selected but libkaiju_gpu.so sees no HIP device (no CPU fallback exists)")
    return api
