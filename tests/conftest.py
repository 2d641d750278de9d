import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # CPU-side checkers are cheap to (re)build; the HIP library normally exists already (__graft_entry__.build()
    # makes it and it travels with the snapshot) -- compile it only if it is missing (hipcc, a few minutes)
    for sub in ("oracle", os.path.join("tests", "cpp")):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, sub)], check=True, stdout=subprocess.DEVNULL)
    if not os.path.exists(os.path.join(ROOT, "brdf_amd", "libbrdf_hip.so")):
        subprocess.run(["make", "-s", "-j", "4", "-C", os.path.join(ROOT, "brdf_amd", "csrc")], check=True)


@pytest.fixture(scope="session")
def libs():
    from tests import oracle_libs
    return oracle_libs
