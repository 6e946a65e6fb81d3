import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "timeout: per-test limit (pytest-timeout)")


def pytest_collection_modifyitems(config, items):
    """A GPU test that never returns (a wedged queue, a collective that never completes) must end the run with a stack dump, not sit
    until the box's own limit: every gpu test gets a 300 s pytest-timeout limit (the longest, the full-size C4 shard, takes ~20 s),
    enforced from a watchdog thread — a signal handler cannot run while the main thread is inside a C call that does not return."""
    try:
        import pytest_timeout  # noqa: F401
    except ImportError:
        return
    for item in items:
        if item.get_closest_marker("gpu") is not None and item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(300, method="thread"))


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle

    oracle.build()
    return oracle
