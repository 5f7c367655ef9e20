import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import rt_loader  # noqa: E402

rt_loader.load()

GOLDEN = os.path.join(ROOT, "tests", "golden")
MODELS = os.path.join(ROOT, "assets", "models")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def models_dir():
    return MODELS
