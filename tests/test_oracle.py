"""CPU tests of the oracle itself: frozen golden vectors, known answers, reference-captured fixtures.

Nothing here touches the GPU or the HIP library.  The oracle (``oracle/cpu_ref.py``) is test
infrastructure; these tests are what it is pinned by (parity with the reference's own arithmetic
is UNPINNED: the reference holds no golden vector for this path, see the oracle header).
"""

import numpy as np
import pytest

from oracle import cpu_ref as o

DESKEW_CASES = ["deskew_nooverhang_avg3", "deskew_overhang_avg1", "deskew_nooverhang_avg5_r0p4"]


@pytest.mark.parametrize("name", DESKEW_CASES)
def test_deskew_matches_frozen_golden(golden_dir, name):
    g = np.load(golden_dir / f"{name}.npz")
    out = o.deskew(g["raw"], float(g["ls_angle_deg"]), float(g["px_to_scan_ratio"]),
                   bool(g["keep_overhang"]), int(g["average_n_slices"]))
    assert out.dtype == np.float32
    np.testing.assert_array_equal(out, g["out"])


def test_affine_matches_frozen_golden(golden_dir):
    g = np.load(golden_dir / "affine_rot2deg.npz")
    np.testing.assert_array_equal(o.affine_apply_4x4(g["vol"], g["matrix"], g["vol"].shape), g["out_constant"])
    out_g = o.affine_apply_4x4(g["vol"], g["matrix"], tuple(g["grid_shape"]), cval=float(g["grid_cval"]),
                               mode="grid-constant")
    np.testing.assert_array_equal(out_g, g["out_grid"])


def test_rl_matches_frozen_golden(golden_dir):
    g = np.load(golden_dir / "rl_5iter.npz")
    np.testing.assert_array_equal(o.richardson_lucy(g["y"], g["psf_sep"], 5), g["x_sep_5"])
    np.testing.assert_array_equal(o.richardson_lucy(g["y"], g["psf_rot"], 5), g["x_rot_5"])


def test_flatfield_matches_reference_capture(golden_dir):
    """The neighbour step: oracle == what the reference's own ``_flat_field_BF`` produced
    (captured by ``oracle/make_golden.py`` importing ``/root/reference/shrimpy/preprocessing.py``)."""
    g = np.load(golden_dir / "ref_preprocessing.npz")
    np.testing.assert_allclose(o.flat_field_bf(g["flatfield_in"]), g["flatfield_out"], rtol=2e-6)
    np.testing.assert_allclose(o.flat_field_bf(g["flatfield_in_odd"]), g["flatfield_out_odd"], rtol=2e-6)


# ---------------------------------------------------------------- known answers


def test_deskew_shape_rule_known_values():
    # config-2 mapping (2048, 512, 2048), 30 deg, r=0.755, avg 3
    shape, voxel = o.deskewed_shape((2048, 512, 2048), 30.0, 0.755, False, 3, 0.1133)
    assert shape == (171, 2048, 2270)
    assert voxel == pytest.approx((3 * 0.5 * 0.1133, 0.1133, 0.1133))
    shape_k, _ = o.deskewed_shape((2048, 512, 2048), 30.0, 0.755, True, 1)
    assert shape_k == (512, 2048, int(np.ceil(2048 / 0.755 + 512 * np.cos(np.pi / 6))))


def test_deskew_reproduces_a_linear_field_exactly_inside():
    """Order-1 interpolation is exact on a field linear in z: the deskew of raw[z,y,x] = z is the
    z-coordinate map itself wherever the sample is in range."""
    Z, Y, X = 40, 10, 6
    raw = np.broadcast_to(np.arange(Z, dtype=np.float32)[:, None, None], (Z, Y, X)).copy()
    m, off, shape = o.deskew_geometry(raw.shape, 30.0, 0.755, True)
    out = o.affine_apply(raw, m, off, shape)
    zp, _, xp = np.meshgrid(*[np.arange(n) for n in shape], indexing="ij")
    z_in = m[0, 0] * zp + m[0, 2] * xp + off[0]
    inside = (z_in >= 0) & (z_in <= Z - 1)
    np.testing.assert_allclose(out[inside], z_in[inside], rtol=0, atol=1e-4)
    assert np.all(out[~inside] == 0)


def test_deskew_chunks_along_raw_x_concatenate_reversed():
    """``scripts/measure_psf.py:221,249``: deskew is independent along raw X and chunks are
    concatenated in REVERSE order on output axis -2."""
    rng = np.random.default_rng(5)
    raw = rng.random((40, 12, 16)).astype(np.float32)
    whole = o.deskew(raw, 30.0, 0.755, True, 3)
    chunks = [o.deskew(c, 30.0, 0.755, True, 3) for c in np.split(raw, 4, axis=-1)]
    np.testing.assert_array_equal(np.concatenate(chunks[::-1], axis=-2), whole)


def test_a_deskewed_plane_is_one_tilt_row_across_the_scan_stack():
    """The reference's viewer builds a deskewed plane from "a single tilt row across the whole scan stack"
    (``shrimpy/viewer/ring_buffer.py:98-105``, ``shrimpy/viewer/_napari_process.py:202-217``): before the slice
    averaging, output plane Z' depends on tilt row Y - 1 - Z' of the raw stack and on nothing else -- in the oracle
    and, on a CPU tensor, in the product's host twin."""
    import torch

    from shrimpy_amd.deskew import fast_deskew_zyx

    rng = np.random.default_rng(17)
    raw = rng.poisson(300, (90, 7, 20)).astype(np.float32)
    kw = dict(ls_angle_deg=30.0, px_to_scan_ratio=0.755, keep_overhang=True, average_n_slices=1)
    base = o.deskew(raw, 30.0, 0.755, True, 1)
    assert base.shape[0] == raw.shape[1]
    for row in (0, 3, 6):
        bumped = raw.copy()
        bumped[:, row, :] += 1000.0
        out = o.deskew(bumped, 30.0, 0.755, True, 1)
        changed = [z for z in range(base.shape[0]) if not np.array_equal(out[z], base[z])]
        assert changed == [raw.shape[1] - 1 - row]
        np.testing.assert_array_equal(fast_deskew_zyx(raw_data=torch.as_tensor(bumped), **kw).numpy(), out)


def test_average_slices_edge_padding():
    d = np.arange(5 * 2 * 2, dtype=np.float32).reshape(5, 2, 2)
    a = o.average_slices(d, 3)
    assert a.shape == (2, 2, 2)
    np.testing.assert_array_equal(a[0], (d[0] + d[1] + d[2]) / np.float32(3))
    np.testing.assert_array_equal(a[1], (d[3] + d[4] + d[4]) / np.float32(3))  # edge pad


def test_affine_identity_and_integer_shift():
    rng = np.random.default_rng(7)
    v = rng.random((6, 7, 8)).astype(np.float32)
    np.testing.assert_array_equal(o.affine_apply(v, np.eye(3), np.zeros(3), v.shape), v)
    sh = o.affine_apply(v, np.eye(3), np.array([1.0, 0.0, -2.0]), v.shape)
    np.testing.assert_array_equal(sh[:-1, :, 2:], v[1:, :, :-2])
    assert np.all(sh[-1] == 0) and np.all(sh[:, :, :2] == 0)


def test_constant_mode_has_no_border_blending():
    """with mode="constant" a coordinate at -1e-9 is OUT, n-1 exactly is IN."""
    v = np.ones((4, 4, 4), dtype=np.float32)
    out = o.affine_apply(v, np.eye(3), np.array([-1e-9, 0.0, 0.0]), v.shape)
    assert np.all(out[0] == 0) and np.all(out[1:] == 1)
    out = o.affine_apply(v, np.eye(3), np.array([0.0, 0.0, 0.0]), (4, 4, 4))
    assert np.all(out == 1)  # index n-1 exactly is inside


def test_rl_delta_psf_is_identity():
    rng = np.random.default_rng(9)
    y = (rng.random((6, 8, 10)) * 100 + 1).astype(np.float32)
    psf = np.zeros((3, 3, 3), np.float32)
    psf[1, 1, 1] = 1
    x = o.richardson_lucy(y, psf, 3)
    np.testing.assert_allclose(x, y, rtol=1e-5)


def test_rl_direct_and_fft_agree():
    psf, _ = o.gaussian_psf((5, 5, 5), (1.2, 1.0, 1.0))
    y = o.bead_scene((12, 24, 24), seed=11, psf=psf, density=2e-3)
    a = o.richardson_lucy(y, psf, 5)
    b = o.richardson_lucy(y, psf, 5, use_fft=True)
    np.testing.assert_allclose(a, b, rtol=2e-3, atol=1e-2)


def test_rl_tolerates_all_zero_volume():
    """Autofocus-failed stacks are written as all-zero volumes
    (``shrimpy/tests/test_mantis_integration.py:285-341``)."""
    psf, _ = o.gaussian_psf((3, 3, 3), (1, 1, 1))
    x = o.richardson_lucy(np.zeros((4, 6, 6), np.float32), psf, 4)
    assert np.all(x == 0) and np.all(np.isfinite(x))


def test_rl_increases_bead_contrast_and_stays_nonnegative():
    psf, _ = o.gaussian_psf((9, 7, 7), (2.0, 1.2, 1.2))
    y = o.bead_scene((20, 32, 32), seed=13, psf=psf, density=1e-3)
    x = o.richardson_lucy(y, psf, 20)
    assert x.min() >= 0
    assert x.max() > 1.5 * y.max()
    assert np.all(np.isfinite(x))


def test_even_psf_axis_padding_is_equivalent():
    from scipy import ndimage

    rng = np.random.default_rng(15)
    v = rng.random((6, 9, 9)).astype(np.float32)
    w = rng.random((2, 4, 3)).astype(np.float32)
    np.testing.assert_allclose(
        ndimage.correlate(v, o._as_odd_psf(w), mode="constant"),
        ndimage.correlate(v, w, mode="constant"), rtol=1e-6)


def test_gaussian_psf_is_separable_and_normalised():
    psf, (kz, ky, kx) = o.gaussian_psf()
    assert psf.shape == (9, 7, 7)
    assert abs(psf.sum(dtype=np.float64) - 1) < 1e-6
    np.testing.assert_allclose(psf, kz[:, None, None] * ky[None, :, None] * kx[None, None, :], rtol=1e-6)


def test_separable_rl_matches_dense_rl_for_asymmetric_factors():
    """The CPU-baseline formulation (three correlate1d passes, flipped taps for H) is the same
    algorithm as the dense loop, also when the factors are not symmetric."""
    rng = np.random.default_rng(17)
    ks = [rng.random(n).astype(np.float32) + 0.1 for n in (5, 3, 7)]
    ks = [k / k.sum() for k in ks]
    psf = (ks[0][:, None, None] * ks[1][None, :, None] * ks[2][None, None, :]).astype(np.float32)
    y = (rng.random((9, 12, 15)) * 50 + 1).astype(np.float32)
    np.testing.assert_allclose(o.richardson_lucy_separable(y, ks, 4), o.richardson_lucy(y, psf, 4),
                               rtol=2e-5)


def test_dynatrack_oracle_matches_reference_capture(golden_dir):
    """The f-3 row's oracle against what the reference's own estimator functions produced."""
    g = np.load(golden_dir / "ref_dynatrack.npz")
    a, b, thin = g["a"], g["b"], g["thin"]
    for p, want in zip(g["percentile_p"], g["percentile_a"]):
        assert o.dt_percentile(a, float(p)) == pytest.approx(float(want), rel=1e-6)
    np.testing.assert_allclose(o.dt_intensity_center_of_mass(a), g["icom_a_bg0"], atol=2e-4)
    np.testing.assert_allclose(o.dt_intensity_center_of_mass(a, 300.0), g["icom_a_bg300"], atol=2e-4)
    np.testing.assert_allclose(o.dt_intensity_center_of_mass(np.zeros((4, 5, 6))), g["icom_blank"])
    np.testing.assert_allclose(o.dt_gaussian_blur_3d(a, 1.0), g["blur_a_s1"], rtol=2e-6, atol=1e-4)
    np.testing.assert_allclose(o.dt_gaussian_blur_3d(a, 2.5), g["blur_a_s2p5"], rtol=2e-6, atol=1e-4)
    np.testing.assert_allclose(o.dt_gaussian_blur_3d(thin, 2.0), g["blur_thin_s2"], rtol=2e-6, atol=1e-4)
    blur = o.dt_gaussian_blur_3d((a - a.min()) / (a.max() - a.min()), 2.0)
    assert [o.dt_multiotsu_threshold(blur, c) for c in (0, 1)] == pytest.approx(g["otsu_blur_a"].tolist(), rel=1e-5)
    mask = o.dt_binary_mask(a, sigma=2.0)
    assert (mask != g["mask_a_s2"]).sum() <= 2            # voxels within 1e-7 of the threshold
    np.testing.assert_allclose(o.dt_center_of_mass(g["mask_a_s2"]), g["com_mask_a_s2"], atol=1e-4)
    np.testing.assert_allclose(o.dt_center_of_mass(np.zeros((3, 4, 5), bool)), g["com_empty"])
    np.testing.assert_allclose(o.dt_roi_shift(a), g["roi_shift_a"], atol=2e-4)
    np.testing.assert_allclose(o.dt_roi_shift(a, 50.0), g["roi_shift_a_p50"], atol=2e-4)
    np.testing.assert_allclose(o.dt_roi_shift(b, 90.0, 1.5), g["roi_shift_b_p90_blur"], atol=1e-3)
    np.testing.assert_allclose(o.dt_multiotsu_center_of_mass(a, b, 2.0, 0), g["motsu_shift_ab_s2"], atol=2e-2)
    np.testing.assert_allclose(o.dt_multiotsu_center_of_mass(a, b, 2.0, 1), g["motsu_shift_ab_s2_c1"], atol=2e-2)


def test_dynatrack_pcc_oracle_on_the_references_2d_known_answers():
    """``shrimpy/tests/test_dynatrack.py:85-100``: (32, 32) against itself -> (0, 0); (64, 64) rolled by
    (3, -5) -> (3, -5)."""
    img = np.random.default_rng(42).random((32, 32)).astype(np.float32)
    assert o.dt_phase_cross_corr(img, img.copy()) == (0, 0)
    ref = np.random.default_rng(42).random((64, 64)).astype(np.float32)
    assert o.dt_phase_cross_corr(ref, np.roll(ref, (3, -5), axis=(0, 1))) == (3, -5)


def test_dynatrack_pcc_oracle_matches_reference_capture(golden_dir):
    g = np.load(golden_dir / "ref_dynatrack.npz")
    assert o.dt_phase_cross_corr(g["pcc_ref"], g["pcc_mov"]) == tuple(g["pcc_shift"]) == (1, 2, -3)
    assert o.dt_phase_cross_corr(g["a"], g["b"]) == tuple(g["pcc_shift_ab"])
    assert o.dt_phase_cross_corr(g["pcc_odd"], g["pcc_odd_mov"]) == tuple(g["pcc_shift_odd"])
    assert o.dt_phase_cross_corr(g["pcc_odd"], g["pcc_odd_mov"], 0.5) == tuple(g["pcc_shift_odd_half"])
    np.testing.assert_array_equal(o.dt_match_shape(g["pcc_odd"], (8, 36, 50)), g["match_shape_odd_pad"])
    np.testing.assert_array_equal(o.dt_match_shape(g["pcc_odd"], (4, 36, 25)), g["match_shape_odd_mixed"])


def test_dynatrack_dispatcher_oracle_matches_reference_capture(golden_dir):
    """``dt_compute_shift`` / ``dt_limit_shifts_zyx`` against the reference's own
    ``DynaTrackUpdater._compute_shift`` and ``_limit_shifts_zyx`` outputs (captured fixture)."""
    g = np.load(golden_dir / "ref_dynatrack.npz")
    limits = {"z": (0.5, 2.0), "y": (0.1, 100.0), "x": (0.2, 0.9)}
    for vin, vout in zip(g["limit_shifts_in"], g["limit_shifts_out"]):
        np.testing.assert_array_equal(o.dt_limit_shifts_zyx(vin, limits), vout)
    methods = ("pcc", "intensity_center_of_mass", "roi_center_pcc", "multiotsu_center_of_mass", "multiotsu_pcc")
    kw = dict(scale_z=0.17, scale_yx=0.1133, otsu_sigma=2.0, blob_sigma=4.0, background_percentile=50.0, blur_sigma=1.5)
    for variant, extra in (("plain", {}), ("limited", dict(limits=limits, dampening=(0.5, 1.0, 0.8)))):
        for method, want in zip(methods, g[f"compute_shift_{variant}_xyz_um"]):
            if method in ("roi_center_pcc", "multiotsu_pcc"):
                continue  # (their estimators are covered by the PCC and mask fixtures)
            np.testing.assert_allclose(o.dt_compute_shift(g["a"], g["b"], method, **kw, **extra), want,
                                       rtol=0, atol=2e-4)
