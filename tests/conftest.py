"""pytest configuration: ``gpu`` marker, repo root on sys.path, single-threaded OpenMP.

``OMP_NUM_THREADS=1`` follows the reference's conftest (``shrimpy/tests/conftest.py:11-17``):
torch and other native OpenMP runtimes must not collide inside one process.
"""

import os
import sys

from pathlib import Path

os.environ.setdefault("OMP_NUM_THREADS", "1")

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

import pytest  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"


@pytest.fixture(scope="session")
def device():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")
