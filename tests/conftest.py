import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """The suites need libtrexhip.so and the oracle library: build them (seconds, cross-compiles without a GPU) when the tree
    is fresh, e.g. when the tests run before __graft_entry__.build()."""
    if not (os.path.exists(os.path.join(ROOT, "trex_amd", "libtrexhip.so")) and os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so"))):
        import __graft_entry__ as entry
        entry.build()
    yield
