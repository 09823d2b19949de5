import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the host library and the oracle once per session (seconds); the HIP library is
    built by __graft_entry__.build() and only needed by -m gpu tests."""
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"])
    host_lib = os.path.join(ROOT, "dynearthsol_amd", "libdes_host.so")
    if not os.path.exists(host_lib):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "dynearthsol_amd", "csrc"), "../libdes_host.so"])
