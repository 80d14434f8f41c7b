"""the DynaTrack estimators as HIP kernels (``shrimpy_amd.dynatrack``) against
(i) what the reference's own functions produced (``tests/golden/ref_dynatrack.npz``, captured by
``oracle/make_golden.py`` from /root/reference/shrimpy/dynatrack/tracking.py) and (ii) the numpy
oracle on larger volumes."""

from __future__ import annotations

import sys

from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import cpu_ref as o  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def device():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def golden():
    return np.load(Path(__file__).resolve().parent / "golden" / "ref_dynatrack.npz")


def _t(a, device):
    import torch

    return torch.as_tensor(np.ascontiguousarray(a), device=device)


def test_percentile_and_histogram_match_the_reference(device, golden):
    from shrimpy_amd import dynatrack as d

    a = _t(golden["a"], device)
    for p, want in zip(golden["percentile_p"], golden["percentile_a"]):
        assert d._percentile(a, float(p)) == pytest.approx(float(want), rel=1e-6)
    # the histogram itself is torch.histc bit for bit (integer counts)
    rng = np.random.default_rng(3)
    for vol in (rng.integers(80, 600, (9, 33, 70)).astype(np.float32),
                (rng.standard_normal((20, 64, 130)) * 50).astype(np.float32)):
        lo, hi = d._minmax(_t(vol, device))
        assert (lo, hi) == (float(vol.min()), float(vol.max()))
        for nbins in (256, 17, 1000):
            np.testing.assert_array_equal(d._histc(_t(vol, device), nbins, lo, hi), o.dt_histc(vol, nbins, lo, hi))
    assert d._percentile(_t(np.full((3, 4, 5), 7.0, np.float32), device), 50.0) == 7.0


def test_histogram_in_pieces_adds_up_and_the_entry_refuses_what_its_counters_cannot_hold(device, monkeypatch):
    """The bins are uint32 counters: ``_histc`` histograms a volume of 2**32 voxels or more piece by piece (forced
    here with a small piece), and ``lsr_histogram_f32`` itself refuses such an ``n`` instead of wrapping."""
    import ctypes

    import torch

    from shrimpy_amd import _lib
    from shrimpy_amd import dynatrack as d

    vol = np.random.default_rng(5).integers(80, 600, (7, 33, 70)).astype(np.float32)
    whole = d._histc(_t(vol, device), 64, 80.0, 599.0)
    monkeypatch.setattr(d, "_HIST_PIECE", 1001)
    np.testing.assert_array_equal(d._histc(_t(vol, device), 64, 80.0, 599.0), whole)
    np.testing.assert_array_equal(whole, o.dt_histc(vol, 64, 80.0, 599.0))
    counts = torch.zeros(64, dtype=torch.int32, device=device)
    with pytest.raises(_lib.LsrUnsupported, match="32 bits"):
        _lib.call("lsr_histogram_f32", _t(vol, device).data_ptr(), 1 << 32, ctypes.c_float(0.0), ctypes.c_float(1.0), 64,
                  counts.data_ptr(), None)


def test_intensity_center_of_mass_matches_the_reference(device, golden):
    import torch

    from shrimpy_amd import dynatrack as d

    a, b = _t(golden["a"], device), _t(golden["b"], device)
    np.testing.assert_allclose(d._intensity_center_of_mass(a).cpu().numpy(), golden["icom_a_bg0"], atol=2e-4)
    np.testing.assert_allclose(d._intensity_center_of_mass(a, 300.0).cpu().numpy(), golden["icom_a_bg300"], atol=2e-4)
    blank = d._intensity_center_of_mass(torch.zeros(4, 5, 6, device=device))
    np.testing.assert_array_equal(blank.cpu().numpy(), golden["icom_blank"])
    np.testing.assert_allclose(d._intensity_center_of_mass_to_roi_center(a), golden["roi_shift_a"], atol=2e-4)
    np.testing.assert_allclose(d._intensity_center_of_mass_to_roi_center(a, background_percentile=50.0),
                               golden["roi_shift_a_p50"], atol=2e-4)
    np.testing.assert_allclose(d._intensity_center_of_mass_to_roi_center(b, background_percentile=90.0, blur_sigma=1.5),
                               golden["roi_shift_b_p90_blur"], atol=1e-3)


def test_gaussian_blur_matches_the_reference(device, golden):
    from shrimpy_amd import dynatrack as d

    a, thin = _t(golden["a"], device), _t(golden["thin"], device)
    np.testing.assert_allclose(d._gaussian_blur_3d(a, 1.0).cpu().numpy(), golden["blur_a_s1"], rtol=2e-6, atol=1e-4)
    np.testing.assert_allclose(d._gaussian_blur_3d(a, 2.5).cpu().numpy(), golden["blur_a_s2p5"], rtol=2e-6, atol=1e-4)
    np.testing.assert_allclose(d._gaussian_blur_3d(thin, 2.0).cpu().numpy(), golden["blur_thin_s2"], rtol=2e-6, atol=1e-4)
    assert d._gaussian_blur_3d(a, 0.0) is a


def test_multiotsu_mask_and_centroid_match_the_reference(device, golden):
    import torch

    from shrimpy_amd import dynatrack as d

    a, b = _t(golden["a"], device), _t(golden["b"], device)
    an = golden["a"]
    blur = d._gaussian_blur_3d(_t((an - an.min()) / (an.max() - an.min()), device), 2.0)
    got = [d._multiotsu_threshold(blur, c) for c in (0, 1)]
    assert got == pytest.approx(golden["otsu_blur_a"].tolist(), rel=1e-5)
    mask = d._binary_mask(a, sigma=2.0, otsu_component=0)
    assert mask.dtype == torch.bool and int((mask.cpu().numpy() != golden["mask_a_s2"]).sum()) <= 2
    np.testing.assert_allclose(d._center_of_mass(_t(golden["mask_a_s2"], device)).cpu().numpy(),
                               golden["com_mask_a_s2"], atol=1e-4)
    np.testing.assert_array_equal(d._center_of_mass(torch.zeros(3, 4, 5, dtype=torch.bool, device=device)).cpu().numpy(),
                                  golden["com_empty"])
    np.testing.assert_allclose(d._multiotsu_center_of_mass(a, b, sigma=2.0, otsu_component=0),
                               golden["motsu_shift_ab_s2"], atol=2e-2)
    np.testing.assert_allclose(d._multiotsu_center_of_mass(a, b, sigma=2.0, otsu_component=1),
                               golden["motsu_shift_ab_s2_c1"], atol=2e-2)
    flat = torch.full((4, 6, 8), 3.0, device=device)
    assert not bool(d._binary_mask(flat).any())


def test_phase_cross_correlation_matches_the_reference(device, golden):
    """Shifts are integers: equality with what the reference returned (its own test case
    rng(42) rolled by (1, 2, -3) included), odd shapes, a cropping maximum_shift, and the two
    methods built on it."""
    from shrimpy_amd import dynatrack as d

    g = golden
    a, b = _t(g["a"], device), _t(g["b"], device)
    assert d._phase_cross_corr(_t(g["pcc_ref"], device), _t(g["pcc_mov"], device)) == (1, 2, -3)
    assert d._phase_cross_corr(a, b) == tuple(g["pcc_shift_ab"])
    odd, odd_mov = _t(g["pcc_odd"], device), _t(g["pcc_odd_mov"], device)
    assert d._phase_cross_corr(odd, odd_mov) == tuple(g["pcc_shift_odd"])
    assert d._phase_cross_corr(odd, odd_mov, 0.5) == tuple(g["pcc_shift_odd_half"])
    np.testing.assert_array_equal(d._match_shape(odd, (8, 36, 50)).cpu().numpy(), g["match_shape_odd_pad"])
    np.testing.assert_array_equal(d._match_shape(odd, (4, 36, 25)).cpu().numpy(), g["match_shape_odd_mixed"])
    assert d._roi_center_pcc(b, blob_sigma=4.0) == tuple(g["roi_pcc_b"])
    assert d._multiotsu_pcc(a, b, sigma=2.0) == tuple(g["motsu_pcc_ab"])
    assert [d._next_fast_len(n) for n in (1, 7, 11, 49, 171, 2270)] == [o.dt_next_fast_len(n) for n in (1, 7, 11, 49, 171, 2270)]


def test_phase_cross_correlation_of_2d_images_as_the_reference_tests_it(device):
    """The reference's own 2-D known answers (``shrimpy/tests/test_dynatrack.py:85-100``): rng(42) (32, 32)
    against itself -> (0, 0); rng(42) (64, 64) rolled by (3, -5) -> (3, -5); plus odd sizes vs the oracle."""
    import torch

    from shrimpy_amd import dynatrack as d

    img = _t(np.random.default_rng(42).random((32, 32)).astype(np.float32), device)
    assert d._phase_cross_corr(img, img.clone()) == (0, 0)
    ref = _t(np.random.default_rng(42).random((64, 64)).astype(np.float32), device)
    assert d._phase_cross_corr(ref, torch.roll(ref, shifts=(3, -5), dims=(0, 1))) == (3, -5)
    rng = np.random.default_rng(9)
    a = rng.random((45, 77)).astype(np.float32)
    b = np.roll(a, (-4, 9), axis=(0, 1))
    assert d._phase_cross_corr(_t(a, device), _t(b, device)) == o.dt_phase_cross_corr(a, b) == (-4, 9)
    assert d._phase_cross_corr(_t(a, device), _t(b, device), 0.6) == o.dt_phase_cross_corr(a, b, 0.6)
    with pytest.raises(ValueError, match="two \\(Y, X\\) images or two"):
        d._phase_cross_corr(_t(a, device), _t(a[None], device))


def test_phase_cross_correlation_on_a_larger_volume(device):
    from oracle.make_golden import dynatrack_scene
    from shrimpy_amd import dynatrack as d

    ref = dynatrack_scene(5, shape=(40, 130, 300))
    mov = np.roll(ref, (3, -11, 25), axis=(0, 1, 2))
    assert d._phase_cross_corr(_t(ref, device), _t(mov, device)) == o.dt_phase_cross_corr(ref, mov) == (3, -11, 25)


@pytest.mark.parametrize("shape,sigma", [((40, 130, 300), 5.0), ((7, 300, 65), 3.0), ((171, 96, 257), 1.0)])
def test_estimators_vs_oracle_on_larger_volumes(device, shape, sigma):
    """Every axis longer / shorter than the 64-long blur segments and the 41-tap kernel."""
    from oracle.make_golden import dynatrack_scene
    from shrimpy_amd import dynatrack as d

    vol = dynatrack_scene(sum(shape), shape=shape)
    t = _t(vol, device)
    np.testing.assert_allclose(d._gaussian_blur_3d(t, sigma).cpu().numpy(), o.dt_gaussian_blur_3d(vol, sigma),
                               rtol=3e-6, atol=2e-4)
    assert d._percentile(t, 75.0) == pytest.approx(o.dt_percentile(vol, 75.0), rel=1e-6)
    np.testing.assert_allclose(d._intensity_center_of_mass(t, 150.0).cpu().numpy(),
                               o.dt_intensity_center_of_mass(vol, 150.0), atol=1e-3)
    np.testing.assert_allclose(d._intensity_center_of_mass_to_roi_center(t, 60.0, sigma), o.dt_roi_shift(vol, 60.0, sigma),
                               atol=2e-3)
    other = dynatrack_scene(sum(shape) + 1, shape=shape, shift=(0.5, 2.0, -3.0))
    np.testing.assert_allclose(d._multiotsu_center_of_mass(t, _t(other, device), sigma=sigma),
                               o.dt_multiotsu_center_of_mass(vol, other, sigma), atol=5e-2)


def test_cpu_tensors_are_never_moved_to_the_device_silently(device):
    """A CPU tensor is processed where it lives (host twins), with one RuntimeWarning per process on a box that has a
    HIP device; the result stays on the CPU.  (Twins == kernels: tests/test_dynatrack_host.py.)"""
    import warnings

    import torch

    from shrimpy_amd import dynatrack as d
    from shrimpy_amd import host

    vol = torch.arange(24, dtype=torch.float32).reshape(2, 3, 4)
    host._warned[0] = False
    with pytest.warns(RuntimeWarning, match="CPU tensor"):
        assert d._percentile(vol, 50.0) == d._percentile(vol.to(device), 50.0)
    with warnings.catch_warnings():
        warnings.simplefilter("error")          # ... once per process
        assert d._gaussian_blur_3d(vol, 1.0).device.type == "cpu"


def test_phase_cross_corr_reuses_the_reference_spectrum_safely(device):
    """Same reference tensor across timepoints: one forward FFT is skipped, the shifts do not change;
    an in-place edit of the reference or another tensor is a miss."""
    import torch

    from shrimpy_amd import dynatrack as d

    rng = np.random.default_rng(77)
    ref = torch.as_tensor(rng.random((12, 40, 36)).astype(np.float32), device=device)
    d.set_spectrum_cache_bytes(1 << 30)
    d._spectra.hits = d._spectra.misses = 0
    want = []
    for k, shift in enumerate(((1, 2, -3), (0, -4, 5), (-2, 0, 1))):
        mov = torch.roll(ref, shifts=shift, dims=(0, 1, 2))
        got = d._phase_cross_corr(ref, mov)
        want.append(got)
        assert got == shift
    assert (d._spectra.hits, d._spectra.misses) == (2, 1)
    d.set_spectrum_cache_bytes(0)                      # off: same answers without the cache
    for shift, w in zip(((1, 2, -3), (0, -4, 5), (-2, 0, 1)), want):
        assert d._phase_cross_corr(ref, torch.roll(ref, shifts=shift, dims=(0, 1, 2))) == w
    d.set_spectrum_cache_bytes(1 << 30)
    d._spectra.hits = d._spectra.misses = 0
    d._phase_cross_corr(ref, ref)
    ref.mul_(2.0).add_(torch.roll(ref, 1, 0))          # in-place edit bumps the version: a miss
    assert d._phase_cross_corr(ref, torch.roll(ref, shifts=(1, 1, 1), dims=(0, 1, 2))) == (1, 1, 1)
    assert d._spectra.hits == 0 and d._spectra.misses == 2
    other = ref.clone()
    assert d._phase_cross_corr(other, torch.roll(other, shifts=(0, 2, 0), dims=(0, 1, 2))) == (0, 2, 0)
    assert d._spectra.misses == 3
    n_before = len(d._spectra._entries)
    del other
    import gc
    gc.collect()
    assert len(d._spectra._entries) == n_before - 1   # the entry went with its tensor
    d.set_spectrum_cache_bytes(8 << 30)


def test_reference_results_are_reused_by_every_two_volume_tracker(device):
    """multiotsu centre / mask of the stored reference and the ROI blob are computed once; the
    shifts equal the uncached ones."""
    import torch

    from shrimpy_amd import dynatrack as d

    rng = np.random.default_rng(9)
    base = np.zeros((24, 72, 80), np.float32)
    base[8:15, 20:44, 30:58] = 50.0
    ref = torch.as_tensor(base + rng.random(base.shape).astype(np.float32), device=device)
    movs = [torch.roll(ref, shifts=s, dims=(0, 1, 2)) for s in ((1, 3, -2), (0, -5, 4))]
    d.set_spectrum_cache_bytes(0)
    want = [(d._multiotsu_center_of_mass(ref, m, 2.0), d._multiotsu_pcc(ref, m, 2.0), d._roi_center_pcc(m, 6.0))
            for m in movs]
    d.set_spectrum_cache_bytes(1 << 30)
    d._spectra.hits = d._spectra.misses = 0
    got = [(d._multiotsu_center_of_mass(ref, m, 2.0), d._multiotsu_pcc(ref, m, 2.0), d._roi_center_pcc(m, 6.0))
           for m in movs]
    assert got == want
    # second timepoint: centre, mask, the mask's spectrum and the blob's spectrum all hit
    assert d._spectra.hits == 4
    d.set_spectrum_cache_bytes(8 << 30)


@pytest.mark.parametrize("shape,axis,r", [
    ((40, 70, 300), 0, 3), ((40, 70, 300), 1, 12), ((40, 70, 300), 1, 13), ((40, 70, 300), 2, 12),
    ((1000, 5, 260), 0, 8),      # marching form, two segments along the axis
    ((3, 1030, 129), 1, 5),      # marching form, segments and a partial column strip
    ((33, 2, 7), 0, 0), ((33, 2, 7), 0, 12), ((20, 9, 11), 0, 12),
    ((90, 6, 134), 0, 20), ((5, 150, 260), 1, 20), ((70, 4, 131), 0, 20),   # packed two-column form / odd inner
])
def test_blur_axis_kernels_match_a_mirror_correlate(device, shape, axis, r):
    """Every form of the one-axis reflect blur (tiled, marching, contiguous) against
    ``scipy.ndimage.correlate1d(mode="mirror")`` -- F.pad's "reflect" -- with the [0, 1] map fused."""
    import ctypes

    import torch
    from scipy import ndimage

    from shrimpy_amd import dynatrack as d

    rng = np.random.default_rng(sum(shape) + 7 * axis + r)
    vol = (rng.random(shape) * 900 + 100).astype(np.float32)
    taps = rng.random(2 * r + 1).astype(np.float32)
    taps /= taps.sum()
    sub, div = float(vol.min()), float(vol.max() - vol.min())
    want = ndimage.correlate1d(((vol - np.float32(sub)) / np.float32(div)).astype(np.float64), taps.astype(np.float64),
                               axis=axis, mode="mirror")
    src, out = _t(vol, device), torch.empty(shape, dtype=torch.float32, device=device)
    dt = _t(taps, device)
    d._run(device, "lsr_blur_reflect_f32", src.data_ptr(), out.data_ptr(), *shape, axis, dt.data_ptr(), r,
           ctypes.c_float(sub), ctypes.c_float(div))
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=2e-6, atol=2e-6)


def test_compute_shift_dispatcher_matches_the_reference_updater(device, golden):
    """Every tracking method through ``compute_shift`` (method dispatch, pixels -> microns, limits,
    dampening, axis order) against ``DynaTrackUpdater._compute_shift`` run on the same volumes."""
    from shrimpy_amd import dynatrack as d

    a, b = _t(golden["a"], device), _t(golden["b"], device)
    limits = {"z": (0.5, 2.0), "y": (0.1, 100.0), "x": (0.2, 0.9)}
    for vin, vout in zip(golden["limit_shifts_in"], golden["limit_shifts_out"]):
        np.testing.assert_array_equal(d._limit_shifts_zyx(vin, limits), vout)
    common = dict(segmentation=dict(otsu_sigma=2.0, otsu_component=0), scale_z=0.17, scale_yx=0.1133,
                  roi_center=dict(blob_sigma=4.0, background_percentile=50.0, blur_sigma=1.5))
    for variant, shift in (("plain", dict(maximum=1.0)),
                           ("limited", dict(maximum=1.0, limits=limits, dampening=(0.5, 1.0, 0.8)))):
        for method, want in zip(d.TRACKING_METHODS, golden[f"compute_shift_{variant}_xyz_um"]):
            got = d.compute_shift(a, b, method, shift=shift, **common)
            np.testing.assert_allclose(got, want, rtol=0, atol=2e-4, err_msg=f"{variant} {method}")
    with pytest.raises(ValueError, match="Unknown tracking_method"):
        d.compute_shift(a, b, "nope")
    with pytest.raises(Exception):
        d.ShiftSettings(maximum=1.0, typo=1)


def test_reference_cache_sees_kernels_that_write_through_out(device):
    """A reference refreshed IN PLACE by one of this package's kernels (``out=``) must be a cache miss:
    the kernels write through raw pointers, so the entry points bump torch's version counter."""
    import torch

    from shrimpy_amd import dynatrack as d
    from shrimpy_amd.register import apply_affine_transform_zyx

    rng = np.random.default_rng(21)
    a = torch.as_tensor(rng.random((8, 32, 32)).astype(np.float32), device=device)
    ref = a.clone()
    mov = torch.roll(a, (1, 2, -3), dims=(0, 1, 2))
    assert d._phase_cross_corr(ref, mov) == (1, 2, -3)
    hits = d._spectra.hits
    assert d._phase_cross_corr(ref, mov) == (1, 2, -3) and d._spectra.hits == hits + 1
    # refresh the reference buffer in place with a shifted volume, through the affine kernel's out=
    shift = np.eye(4)
    shift[:3, 3] = [0.0, 0.0, 1.0]
    v0 = ref._version
    apply_affine_transform_zyx(mov.clone(), shift, out=ref)
    assert ref._version > v0
    want = d._phase_cross_corr(ref.clone(), mov)           # an uncached tensor with the same content
    misses = d._spectra.misses
    assert d._phase_cross_corr(ref, mov) == want and d._spectra.misses == misses + 1
    d.invalidate_reference(ref)
    misses = d._spectra.misses
    assert d._phase_cross_corr(ref, mov) == want and d._spectra.misses == misses + 1
