import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The suite runs on the -DCLC_TEST_HOOKS build: the same translation units as the product library csrc/libclc_hip.so plus the
# clc_debug_* / clc_time_* hooks the tests drive and inspect the paths with (which the product library does not export:
# tests/test_abi_symbols.py).  tests/test_gpu_product_library.py runs the product build itself and compares bit for bit.
# CLC_LIBRARY set by the caller (another build of the same units) wins.
_HOOKS = os.path.join(ROOT, "camlasercalibratool_amd", "csrc", "libclc_hip_hooks.so")
if "CLC_LIBRARY" not in os.environ and os.path.exists(_HOOKS):
    os.environ["CLC_LIBRARY"] = _HOOKS


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "timeout: per-test limit (pytest-timeout)")


def pytest_collection_modifyitems(config, items):
    """A GPU test that never returns (a wedged queue, a collective that never completes) must end the run with a stack dump, not sit
    until the box's own limit: every gpu test gets a 600 s pytest-timeout limit (the longest, the full-size C4 shard, takes ~20 s),
    enforced from a watchdog thread — a signal handler cannot run while the main thread is inside a C call that does not return."""
    try:
        import pytest_timeout  # noqa: F401
    except ImportError:
        return
    for item in items:
        if item.get_closest_marker("gpu") is not None and item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(600, method="thread"))


def pytest_collection_finish(session):
    """On a fresh GPU box the first `import torch` pages the wheel in from a cold image — observed: more than five minutes once, inside
    the first test that needed torch.distributed, whose per-test limit then ended the whole `-x` run.  When GPU tests are selected the
    import happens here, before any test's clock runs."""
    if any(item.get_closest_marker("gpu") is not None for item in session.items):
        try:
            import torch.distributed  # noqa: F401
        except ImportError:
            pass


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle

    oracle.build()
    return oracle
