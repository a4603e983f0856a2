import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the tests drive A/B switches, micro-benchmarks and probes: they (and every subprocess they start) run the LL_TUNING=1 build of the
# library (libllamole_hip_tuning.so; llamole_amd/_lib.py).  tests/test_abi_cpu.py checks the product library on its own.
os.environ.setdefault("LLAMOLE_TUNING", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if os.environ.get("LLAMOLE_FILL_UNINIT") == "1":
        # hunt for reliance on what torch.empty happens to return: every uninitialised allocation is filled (NaN / the largest integer), so
        # a kernel or wrapper that assumes zeros fails on the first run instead of once the caching allocator hands back used memory
        import torch
        torch.use_deterministic_algorithms(True, warn_only=True)
        torch.utils.deterministic.fill_uninitialized_memory = True


@pytest.fixture(scope="session")
def hip_available():
    import torch
    return torch.cuda.is_available()


@pytest.fixture(autouse=True)
def _engine_buffer_guards():
    """LL_DEBUG_POISON=1 (the out-of-bounds hunt): after every test, no kernel may have written past the end of an engine buffer."""
    yield
    if os.environ.get("LL_DEBUG_POISON"):
        import torch
        if torch.cuda.is_available():
            from llamole_amd import _lib
            assert _lib.load().ll_debug_check_guards() == 0, "a kernel wrote past the end of an engine buffer (see LL_GUARD on stderr)"
