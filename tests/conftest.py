import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference mounted (build container only)")


def _gpu_present() -> bool:
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    markexpr = config.getoption("-m") or ""
    for item in items:
        if "gpu" in item.keywords and "gpu" not in markexpr.replace("not gpu", ""):
            if not _gpu_present():
                item.add_marker(pytest.mark.skip(reason="no GPU in this container"))


@pytest.fixture(scope="session")
def ctx():
    from padne_amd import solver
    c = solver.get_context()
    yield c


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
