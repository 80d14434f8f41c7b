"""pytest configuration: ``gpu`` marker, repo root on sys.path, single-threaded OpenMP.

``OMP_NUM_THREADS=1`` follows the reference's conftest (``shrimpy/tests/conftest.py:11-17``):
torch and other native OpenMP runtimes must not collide inside one process.
"""

import os
import sys
import time

from pathlib import Path

os.environ.setdefault("OMP_NUM_THREADS", "1")

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

import pytest  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if _TRACE:   # Python stacks of every thread at a fatal signal, beside the trace (never in the log's tail)
        import faulthandler

        global _FAULT_FILE
        _FAULT_FILE = open(_TRACE + ".fault", "w")
        faulthandler.enable(file=_FAULT_FILE, all_threads=True)


_T0 = time.monotonic()
_FAULT_FILE = None
_TRACE = os.environ.get("LSR_TEST_TRACE")  # optional file that receives the same lines


def _announce(line: str) -> None:
    """One flushed line on the real stderr (past pytest's capture), so that the tail of a run
    that dies in native code (abort, GPU fault) ends with the test that was running."""
    line = f"[lsr-test {time.monotonic() - _T0:7.1f}s] {line}"
    try:
        os.write(2, (line + "\n").encode())
    except OSError:
        pass
    if _TRACE:
        with open(_TRACE, "a") as fh:
            fh.write(line + "\n")


def pytest_runtest_logstart(nodeid, location):
    _announce(f"start {nodeid}")


def pytest_runtest_logfinish(nodeid, location):
    _announce(f"done  {nodeid}")


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"


@pytest.fixture(scope="session")
def device():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")


def pytest_sessionfinish(session, exitstatus):
    """Mark the end of the tests proper: anything that dies after this line died in teardown (garbage
    collection of device objects, library destructors at interpreter exit), not in a test."""
    _announce(f"session finished, exit status {int(exitstatus)}: {session.testscollected} collected, "
              f"{session.testsfailed} failed")
