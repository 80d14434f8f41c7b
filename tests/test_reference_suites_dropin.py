"""The reference's OWN test files, run in place with this package standing in for what they exercise.

Only where ``/root/reference`` exists (this build container; it cannot travel to the GPU box): the reference's
``shrimpy/tests/test_dynatrack.py`` -- 72 tests of its estimators, trackers and updater -- with every private estimator
of ``shrimpy.dynatrack.tracking`` swapped for ``shrimpy_amd.dynatrack``'s function of the same name, and its
``shrimpy/tests/test_preprocessing.py`` with the absent ``biahub`` bound to this package -- including the test that
compares the reference's flat-field with biahub's (skipped upstream here: no biahub).  Product code on CPU tensors
(the native host twins), the reference's own assertions.  ``tests/_run_reference_suite.py`` is the runner.
"""

import re
import subprocess
import sys

from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
REFERENCE = Path("/root/reference")

pytestmark = pytest.mark.skipif(not (REFERENCE / "shrimpy" / "tests" / "test_dynatrack.py").exists(),
                                reason="the reference checkout is not on this machine")


def _run(which: str):
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "_run_reference_suite.py"), which, str(REFERENCE)],
                       cwd="/tmp", capture_output=True, text=True, timeout=900,
                       env={"PATH": "/usr/bin:/bin", "PYTHONDONTWRITEBYTECODE": "1", "OMP_NUM_THREADS": "1",
                            "HOME": "/tmp"})
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    m = re.search(r"(\d+) passed(?:, (\d+) skipped)?", r.stdout)
    assert m, tail
    return int(m.group(1)), int(m.group(2) or 0), r.stdout


def test_the_references_dynatrack_tests_pass_with_this_packages_estimators():
    passed, skipped, _ = _run("dynatrack")
    assert passed >= 72 and skipped == 0


def test_the_references_preprocessing_tests_pass_with_this_package_bound_as_biahub():
    passed, skipped, out = _run("preprocessing")
    # with biahub.flat_field_correction provided the comparison the reference skips without biahub runs too
    assert passed >= 12 and skipped == 0, out[-1500:]
