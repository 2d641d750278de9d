import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # CPU-side checkers are cheap to (re)build; the HIP library must already exist (build() makes it)
    for sub in ("oracle", os.path.join("tests", "cpp")):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, sub)], check=True, stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def libs():
    from tests import oracle_libs
    return oracle_libs
