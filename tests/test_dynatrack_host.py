"""The DynaTrack estimators on CPU tensors: the native host twins (``csrc/estimators_host.hip``) + ``torch.fft``.

The reference's tracker resolves its device to ``cuda`` if available, else ``cpu`` (``shrimpy/dynatrack/tracking.py:1054``)
and its CI has no GPU, so ``shrimpy_amd.dynatrack`` takes CPU tensors too.  The cases are the ones the device suite
runs -- the fixtures captured from the imported reference (``tests/golden/ref_dynatrack.npz``,
``oracle/make_golden.py``), the oracle on larger volumes, and the reference's own test cases
(``shrimpy/tests/test_dynatrack.py``) -- re-collected here with a CPU ``device``; no GPU is needed, so they run in the
``-m "not gpu"`` suite.  The twins equal the kernels: ``test_host_estimators_equal_the_device_kernels`` (gpu).
"""

import numpy as np
import pytest
import torch

from tests import test_dynatrack_gpu as device_cases
from tests import test_dynatrack_reference_cases_gpu as reference_cases

from shrimpy_amd import dynatrack as d

golden = device_cases.golden      # the fixture, re-collected in this module


@pytest.fixture(scope="module")
def device():
    return torch.device("cpu")


@pytest.fixture(autouse=True)
def _cases_on_the_cpu(monkeypatch):
    monkeypatch.setattr(reference_cases, "DEV", "cpu")
    d.invalidate_reference()


# ---- the fixture-pinned and oracle cases of tests/test_dynatrack_gpu.py
test_percentile_and_histogram_match_the_reference = device_cases.test_percentile_and_histogram_match_the_reference
test_intensity_center_of_mass_matches_the_reference = device_cases.test_intensity_center_of_mass_matches_the_reference
test_gaussian_blur_matches_the_reference = device_cases.test_gaussian_blur_matches_the_reference
test_multiotsu_mask_and_centroid_match_the_reference = device_cases.test_multiotsu_mask_and_centroid_match_the_reference
test_phase_cross_correlation_matches_the_reference = device_cases.test_phase_cross_correlation_matches_the_reference
test_phase_cross_correlation_of_2d_images_as_the_reference_tests_it = (
    device_cases.test_phase_cross_correlation_of_2d_images_as_the_reference_tests_it)
test_phase_cross_correlation_on_a_larger_volume = device_cases.test_phase_cross_correlation_on_a_larger_volume
test_estimators_vs_oracle_on_larger_volumes = device_cases.test_estimators_vs_oracle_on_larger_volumes
test_phase_cross_corr_reuses_the_reference_spectrum_safely = (
    device_cases.test_phase_cross_corr_reuses_the_reference_spectrum_safely)
test_reference_results_are_reused_by_every_two_volume_tracker = (
    device_cases.test_reference_results_are_reused_by_every_two_volume_tracker)
test_blur_axis_kernels_match_a_mirror_correlate = device_cases.test_blur_axis_kernels_match_a_mirror_correlate
test_compute_shift_dispatcher_matches_the_reference_updater = (
    device_cases.test_compute_shift_dispatcher_matches_the_reference_updater)

# ---- the reference's own test cases (tests/test_dynatrack_reference_cases_gpu.py)
test_match_shape_cases_2d = reference_cases.test_match_shape_cases_2d
test_gaussian_blur_cases = reference_cases.test_gaussian_blur_cases
test_binary_mask_cases = reference_cases.test_binary_mask_cases
test_center_of_mass_cases = reference_cases.test_center_of_mass_cases
test_intensity_center_of_mass_cases = reference_cases.test_intensity_center_of_mass_cases
test_percentile_cases = reference_cases.test_percentile_cases
test_roi_centre_shift_cases = reference_cases.test_roi_centre_shift_cases
test_centred_blob_and_roi_centre_pcc_cases = reference_cases.test_centred_blob_and_roi_centre_pcc_cases
test_multiotsu_tracker_cases = reference_cases.test_multiotsu_tracker_cases
test_compute_shift_cases = reference_cases.test_compute_shift_cases


def test_results_do_not_depend_on_the_worker_count():
    """Identical values whatever ``torch.get_num_threads()`` says, except the fp64 centroid sums (range order)."""
    rng = np.random.default_rng(11)
    vol = torch.as_tensor((rng.random((9, 40, 50)) * 500).astype(np.float32))
    other = torch.roll(vol, shifts=(1, -3, 2), dims=(0, 1, 2))
    before = torch.get_num_threads()
    results = []
    try:
        for n in (1, 3, 8):
            torch.set_num_threads(n)
            d.invalidate_reference()
            results.append((d._percentile(vol, 90.0), d._gaussian_blur_3d(vol, 2.0), d._phase_cross_corr(vol, other),
                            d._intensity_center_of_mass(vol, 100.0), d._multiotsu_threshold(d._gaussian_blur_3d(vol, 1.0))))
    finally:
        torch.set_num_threads(before)
    for r in results[1:]:
        assert r[0] == results[0][0] and torch.equal(r[1], results[0][1]) and r[2] == results[0][2] == (1, -3, 2)
        assert r[4] == results[0][4]
        np.testing.assert_allclose(r[3].numpy(), results[0][3].numpy(), rtol=1e-6)


def test_volumes_on_different_devices_are_refused():
    if not torch.cuda.is_available():
        pytest.skip("needs a second device type")
    a = torch.zeros(4, 8, 8)
    with pytest.raises(ValueError, match="is on"):
        d._phase_cross_corr(a, a.cuda())


@pytest.mark.gpu
def test_host_estimators_equal_the_device_kernels():
    """On the GPU box: every estimator gives the same value from a CPU tensor (host twins) as from the device copy."""
    import warnings

    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    vol = torch.as_tensor((rng.random((21, 70, 133)) * 700 + 50).astype(np.float32))
    other = torch.roll(vol, shifts=(2, -4, 7), dims=(0, 1, 2))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)     # "a CPU tensor ... although a HIP device is visible"
        assert d._minmax(vol) == d._minmax(vol.to(dev))
        lo, hi = d._minmax(vol)
        np.testing.assert_array_equal(d._histc(vol, 256, lo, hi), d._histc(vol.to(dev), 256, lo, hi))
        # one blur pass per axis with the SAME taps on both sides: bit-equal (through _gaussian_blur_3d each device
        # computes its own exp(), which may differ in the last bit)
        import ctypes

        for axis, r in ((0, 4), (1, 12), (2, 9), (2, 20)):
            taps = torch.as_tensor(rng.random(2 * r + 1).astype(np.float32))
            out_h, out_d = torch.empty_like(vol), torch.empty_like(vol, device=dev)
            d._run(vol.device, "lsr_blur_reflect_f32", vol.data_ptr(), out_h.data_ptr(), *vol.shape, axis, taps.data_ptr(), r,
                   ctypes.c_float(50.0), ctypes.c_float(700.0))
            vd, td = vol.to(dev), taps.to(dev)
            d._run(dev, "lsr_blur_reflect_f32", vd.data_ptr(), out_d.data_ptr(), *vol.shape, axis, td.data_ptr(), r,
                   ctypes.c_float(50.0), ctypes.c_float(700.0))
            assert torch.equal(out_h, out_d.cpu()), (axis, r)
        for sigma in (1.0, 3.0):
            np.testing.assert_allclose(d._gaussian_blur_3d(vol, sigma).numpy(), d._gaussian_blur_3d(vol.to(dev), sigma).cpu().numpy(),
                                       rtol=2e-6)
        assert torch.equal(d._match_shape(vol, (24, 64, 144)), d._match_shape(vol.to(dev), (24, 64, 144)).cpu())
        np.testing.assert_allclose(d._intensity_center_of_mass(vol, 300.0).numpy(),
                                   d._intensity_center_of_mass(vol.to(dev), 300.0).cpu().numpy(), rtol=1e-6)
        np.testing.assert_allclose(d._center_of_mass(vol, 600.0).numpy(), d._center_of_mass(vol.to(dev), 600.0).cpu().numpy(),
                                   rtol=1e-6)
        assert d._percentile(vol, 99.0) == d._percentile(vol.to(dev), 99.0)
        assert d._phase_cross_corr(vol, other) == d._phase_cross_corr(vol.to(dev), other.to(dev)) == (2, -4, 7)
        assert d._multiotsu_pcc(vol, other, sigma=1.0) == d._multiotsu_pcc(vol.to(dev), other.to(dev), sigma=1.0)
