import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_terminal_summary(terminalreporter):
    """How many GPU / oracle DOA-bin differences the parity bar classified as oracle-fragile in this session, and how many of them
    only under the local scaling of eps (tests/parity_helpers.py) rather than the absolute 1e-6."""
    try:
        import parity_helpers
    except Exception:
        return
    t = parity_helpers.TALLY
    if t["differences_classified"]:
        terminalreporter.write_line("parity bar: %d bin difference(s) classified as oracle-fragile frames; %d under the absolute 1e-6 bar, %d only with "
                                    "eps scaled by the values compared" % (t["differences_classified"], t["of_them_under_the_absolute_bar"],
                                                                            t["of_them_only_under_the_local_bar"]))
