import os
import sys

import pytest
import torch  # noqa: F401  -- before librtamd.so is loaded: torch ships a HIP runtime of its own, and when the
#                              system's is loaded first torch later finds "no HIP GPUs" in the same process

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    config.addinivalue_line("markers", "slow: takes more than a minute on the CPU")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
