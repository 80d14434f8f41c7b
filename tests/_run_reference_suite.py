"""Run one of the REFERENCE's own test files, in place, with this package standing in (not a test module itself).

    python tests/_run_reference_suite.py dynatrack | preprocessing  [/root/reference]

``dynatrack``: ``shrimpy/tests/test_dynatrack.py`` with every private estimator of ``shrimpy.dynatrack.tracking``
replaced by the function of the same name from ``shrimpy_amd.dynatrack`` before the test module is imported.
``preprocessing``: ``shrimpy/tests/test_preprocessing.py`` with ``biahub`` (absent from the image) bound to this
package (``biahub.deskew``, ``biahub.settings``, ``biahub.flat_field_correction``).  Nothing of the reference is
copied or written: its files are imported from where they lie, bytecode writing off, pytest's cache off.
"""
import sys
import types

sys.dont_write_bytecode = True
which = sys.argv[1]
reference = sys.argv[2] if len(sys.argv) > 2 else "/root/reference"
repo = __file__.rsplit("/tests/", 1)[0]
sys.path.insert(0, repo)
sys.path.insert(1, reference)

import pytest  # noqa: E402

ESTIMATORS = ["_gaussian_blur_3d", "_multiotsu_threshold", "_binary_mask", "_center_of_mass", "_percentile",
              "_intensity_center_of_mass", "_intensity_center_of_mass_to_roi_center", "_multiotsu_center_of_mass",
              "_next_fast_len", "_match_shape", "_phase_cross_corr", "_centered_gaussian_blob", "_roi_center_pcc",
              "_multiotsu_pcc"]

if which == "dynatrack":
    import shrimpy.dynatrack.tracking as tracking

    from shrimpy_amd import dynatrack as ours

    swapped = [n for n in ESTIMATORS if hasattr(tracking, n) and hasattr(ours, n)]
    for n in swapped:
        setattr(tracking, n, getattr(ours, n))
    print(f"swapped {len(swapped)} of {len(ESTIMATORS)} estimators", file=sys.stderr)
    if len(swapped) != len(ESTIMATORS):
        sys.exit(3)
    target = f"{reference}/shrimpy/tests/test_dynatrack.py"
elif which == "preprocessing":
    import shrimpy_amd.deskew as our_deskew
    import shrimpy_amd.flatfield as our_flatfield
    import shrimpy_amd.settings as our_settings

    biahub = types.ModuleType("biahub")
    biahub.__path__ = []          # a package, so that importorskip("biahub.flat_field_correction") resolves
    biahub.deskew, biahub.settings, biahub.flat_field_correction = our_deskew, our_settings, our_flatfield
    sys.modules.update({"biahub": biahub, "biahub.deskew": our_deskew, "biahub.settings": our_settings,
                        "biahub.flat_field_correction": our_flatfield})
    target = f"{reference}/shrimpy/tests/test_preprocessing.py"
else:
    sys.exit(f"unknown suite {which!r}")

sys.exit(pytest.main([target, "--noconftest", "-p", "no:cacheprovider", "--rootdir=/tmp", "-c", "/dev/null", "-q", "-rs"]))
