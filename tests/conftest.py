import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionfinish(session, exitstatus):
    from tests import parity_log
    parity_log.dump(ROOT)


@pytest.fixture(scope="session")
def engine():
    """One EvoFst context on cuda:0 for the whole GPU session (contexts are reusable)."""
    from evo_amd import EvoFst
    e = EvoFst(0)
    yield e
    e.close()
