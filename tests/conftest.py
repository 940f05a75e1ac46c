import os
import sys

import pytest

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # before any HIP runtime starts: see mm2-gb_amd/csrc/engine.hip, Engine::init

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
