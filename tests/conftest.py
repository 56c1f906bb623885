import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    from _checkers import Checker
    return Checker("orc")


@pytest.fixture(scope="session")
def ref():
    """The reference TU compiled in place (oracle/_ref); absent => skip."""
    from _checkers import Checker, have_ref
    if not have_ref():
        pytest.skip("oracle/_ref/libtfref.so not built (needs /root/reference)")
    return Checker("ref")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
