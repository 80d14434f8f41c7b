"""Generate the small golden fixtures under tests/golden/ from the CPU oracle.

Run in the build container:  python oracle/make_golden.py
TEST INFRASTRUCTURE ONLY (see oracle/cpu_ref.py header).  The fixtures are DATA: seeded inputs
and the oracle's outputs (scipy 1.15.3 / numpy 2.2.6), so that the GPU box -- which has neither
/root/reference nor any need to re-run scipy for them -- can check the HIP path against frozen
numbers, and so that a change of scipy version shows up as an oracle-vs-golden failure.

A second group of fixtures (``ref_*.npz``) is captured by importing the reference's own
``shrimpy.preprocessing`` from /root/reference (when present): they pin the neighbour step
(flat-field) and the boundary contract, see ``capture_reference_fixtures``.
"""

from __future__ import annotations

import sys

from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import cpu_ref as o  # noqa: E402

GOLD = ROOT / "tests" / "golden"


def deskew_cases():
    rng = np.random.default_rng(1001)
    raw = (rng.random((24, 16, 40)) * 1000).astype(np.float32)
    cases = {}
    for name, (ko, avg, r) in {
        "deskew_nooverhang_avg3": (False, 3, 0.755),
        "deskew_overhang_avg1": (True, 1, 0.755),
        "deskew_nooverhang_avg5_r0p4": (False, 5, 0.4),
    }.items():
        out = o.deskew(raw, 30.0, r, ko, avg)
        shape, voxel = o.deskewed_shape(raw.shape, 30.0, r, ko, avg, 0.1133)
        assert tuple(out.shape) == shape
        cases[name] = dict(raw=raw, out=out, ls_angle_deg=30.0, px_to_scan_ratio=r,
                           keep_overhang=ko, average_n_slices=avg, voxel=np.array(voxel))
    return cases


def affine_case():
    rng = np.random.default_rng(1003)
    vol = (rng.random((12, 40, 48)) * 500).astype(np.float32)
    th = np.deg2rad(2.0)
    rot = np.array([[1, 0, 0], [0, np.cos(th), -np.sin(th)], [0, np.sin(th), np.cos(th)]])
    m = np.eye(4)
    m[:3, :3] = rot @ np.diag([1.0, 0.98, 1.02])
    m[:3, 3] = [0.5, -2.25, 3.75]
    out_c = o.affine_apply_4x4(vol, m, vol.shape)
    out_g = o.affine_apply_4x4(vol, m, (10, 44, 50), cval=7.0, mode="grid-constant")
    return dict(vol=vol, matrix=m, out_constant=out_c, out_grid=out_g, grid_shape=np.array([10, 44, 50]),
                grid_cval=7.0)


def rl_cases():
    rng = np.random.default_rng(1005)
    psf_sep, factors = o.gaussian_psf((5, 5, 5), (1.2, 1.0, 1.0))
    y = o.bead_scene((16, 32, 32), seed=1005, psf=psf_sep, density=2e-3)
    x_sep = o.richardson_lucy(y, psf_sep, iterations=5)
    psf_rot = o.rotated_psf((5, 5, 5), (1.2, 0.8, 1.0), 30.0)
    x_rot = o.richardson_lucy(y, psf_rot, iterations=5)
    ratio1, x1 = o.rl_iteration_parts(y, y, psf_rot)
    return dict(y=y, psf_sep=psf_sep, kz=factors[0], ky=factors[1], kx=factors[2], x_sep_5=x_sep,
                psf_rot=psf_rot, x_rot_5=x_rot, ratio_rot_1=ratio1, x_rot_1=x1,
                norm_rot=o.rl_norm(y.shape, psf_rot), _unused=rng.random(1))


def capture_reference_fixtures():
    """Import the reference's preprocessing module and record what it pins."""
    ref = Path("/root/reference")
    if not ref.exists():
        print("reference absent: ref_*.npz not regenerated")
        return
    sys.path.insert(0, str(ref))
    sys.dont_write_bytecode = True
    import torch

    from shrimpy.preprocessing import RECON_STEPS, _LabelfreePreprocessor, _settings_kwargs

    # (1) flat-field on the reference's own test input (shrimpy/tests/test_preprocessing.py:155-160)
    vol = np.random.default_rng(3).integers(80, 600, (8, 6, 10)).astype(np.float32)
    pre = _LabelfreePreprocessor(zyx_shape=vol.shape, deskew_settings=None, phase_settings=None,
                                 vs_config=None, output_channel="BF", apply_flatfield=True)
    ff = pre._flat_field_BF(torch.as_tensor(vol)).numpy()
    # odd Z too (median picks the middle value)
    vol7 = vol[:7]
    ff7 = pre._flat_field_BF(torch.as_tensor(vol7)).numpy()

    # (2) the channel-dict contract of __call__ with deskew stubbed by an identity
    class _Deskew:
        def model_dump(self):
            return {"ls_angle_deg": 30.0, "px_to_scan_ratio": 0.755, "keep_overhang": False,
                    "average_n_slices": 3, "pixel_size_um": 0.1133, "scan_step_um": 0.15}

    def callee(raw_data, ls_angle_deg, px_to_scan_ratio, keep_overhang, average_n_slices=1):
        return raw_data

    kept = _settings_kwargs(callee, _Deskew())
    np.savez_compressed(
        GOLD / "ref_preprocessing.npz",
        flatfield_in=vol, flatfield_out=ff, flatfield_in_odd=vol7, flatfield_out_odd=ff7,
        recon_steps=np.array(RECON_STEPS),
        settings_kwargs_kept=np.array(sorted(kept)),
    )
    print("captured ref_preprocessing.npz from", ref)


def dynatrack_scene(seed, shape=(12, 40, 48), shift=(0.0, 0.0, 0.0)):
    """A bright blob on a noisy pedestal (the kind of volume DynaTrack centres on)."""
    rng = np.random.default_rng(seed)
    zz, yy, xx = np.meshgrid(*(np.arange(n, dtype=np.float32) for n in shape), indexing="ij")
    c = [(n - 1) / 2 + 0.15 * n + s for n, s in zip(shape, shift)]
    blob = np.exp(-0.5 * (((zz - c[0]) / 2.0) ** 2 + ((yy - c[1]) / 5.0) ** 2 + ((xx - c[2]) / 6.0) ** 2))
    return (100 + 900 * blob + 20 * rng.standard_normal(shape)).astype(np.float32)


def capture_dynatrack_fixtures():
    """Run the reference's own estimator functions (shrimpy/dynatrack/tracking.py) on seeded
    volumes and record inputs + outputs (the importable oracle of the f-3 row)."""
    ref = Path("/root/reference")
    if not ref.exists():
        print("reference absent: ref_dynatrack.npz not regenerated")
        return
    if str(ref) not in sys.path:
        sys.path.insert(0, str(ref))
    sys.dont_write_bytecode = True
    import torch

    from shrimpy.dynatrack import tracking as t

    a = dynatrack_scene(11)
    b = dynatrack_scene(12, shift=(1.0, -3.0, 2.5))
    thin = dynatrack_scene(13, shape=(3, 9, 70))          # axes shorter than the blur radius
    ta, tb, tt = (torch.as_tensor(v) for v in (a, b, thin))
    out = {"a": a, "b": b, "thin": thin}
    out["percentile_a"] = np.array([t._percentile(ta, p) for p in (1.0, 50.0, 90.0, 99.5)])
    out["percentile_p"] = np.array([1.0, 50.0, 90.0, 99.5])
    out["icom_a_bg0"] = t._intensity_center_of_mass(ta).numpy()
    out["icom_a_bg300"] = t._intensity_center_of_mass(ta, background=300.0).numpy()
    out["icom_blank"] = t._intensity_center_of_mass(torch.zeros(4, 5, 6)).numpy()
    out["roi_shift_a"] = np.array(t._intensity_center_of_mass_to_roi_center(ta))
    out["roi_shift_a_p50"] = np.array(t._intensity_center_of_mass_to_roi_center(ta, background_percentile=50.0))
    out["roi_shift_b_p90_blur"] = np.array(
        t._intensity_center_of_mass_to_roi_center(tb, background_percentile=90.0, blur_sigma=1.5))
    out["blur_a_s1"] = t._gaussian_blur_3d(ta, 1.0).numpy()
    out["blur_a_s2p5"] = t._gaussian_blur_3d(ta, 2.5).numpy()
    out["blur_thin_s2"] = t._gaussian_blur_3d(tt, 2.0).numpy()
    blur = t._gaussian_blur_3d((ta - ta.min()) / (ta.max() - ta.min()), 2.0)
    out["otsu_blur_a"] = np.array([t._multiotsu_threshold(blur, 0), t._multiotsu_threshold(blur, 1)])
    mask = t._binary_mask(ta, sigma=2.0, otsu_component=0)
    out["mask_a_s2"] = mask.numpy()
    out["com_mask_a_s2"] = t._center_of_mass(mask).numpy()
    out["com_empty"] = t._center_of_mass(torch.zeros(3, 4, 5, dtype=torch.bool)).numpy()
    out["motsu_shift_ab_s2"] = np.array(t._multiotsu_center_of_mass(ta, tb, sigma=2.0, otsu_component=0))
    out["motsu_shift_ab_s2_c1"] = np.array(t._multiotsu_center_of_mass(ta, tb, sigma=2.0, otsu_component=1))
    # phase cross-correlation (tracking.py:309-378): known rolls, the reference's own test input
    # (tests/test_dynatrack.py:102-111: rng(42).random((8,32,32)) rolled by (1,2,-3)), odd shapes
    pcc_ref = np.random.default_rng(42).random((8, 32, 32)).astype(np.float32)
    pcc_mov = np.roll(pcc_ref, (1, 2, -3), axis=(0, 1, 2))
    out["pcc_ref"], out["pcc_mov"] = pcc_ref, pcc_mov
    out["pcc_shift"] = np.array(t._phase_cross_corr(torch.as_tensor(pcc_ref), torch.as_tensor(pcc_mov)))
    out["pcc_shift_ab"] = np.array(t._phase_cross_corr(ta, tb))
    odd = dynatrack_scene(14, shape=(7, 33, 49))
    odd_mov = np.roll(odd, (-2, 5, 7), axis=(0, 1, 2))
    out["pcc_odd"], out["pcc_odd_mov"] = odd, odd_mov
    out["pcc_shift_odd"] = np.array(t._phase_cross_corr(torch.as_tensor(odd), torch.as_tensor(odd_mov)))
    out["pcc_shift_odd_half"] = np.array(t._phase_cross_corr(torch.as_tensor(odd), torch.as_tensor(odd_mov), 0.5))
    out["match_shape_odd_pad"] = t._match_shape(torch.as_tensor(odd), (8, 36, 50)).numpy()
    out["match_shape_odd_mixed"] = t._match_shape(torch.as_tensor(odd), (4, 36, 25)).numpy()
    out["roi_pcc_b"] = np.array(t._roi_center_pcc(tb, blob_sigma=4.0))
    out["motsu_pcc_ab"] = np.array(t._multiotsu_pcc(ta, tb, sigma=2.0))
    # the dispatcher on top of them (DynaTrackUpdater._compute_shift, tracking.py:1224-1312): method
    # selection, pixels -> microns, limits (deadband / clip), dampening, (z, y, x) -> (x, y, z)
    import types

    methods = ("pcc", "intensity_center_of_mass", "roi_center_pcc", "multiotsu_center_of_mass", "multiotsu_pcc")
    for variant, shift in (("plain", dict(maximum=1.0)),
                           ("limited", dict(maximum=1.0, limits={"z": (0.5, 2.0), "y": (0.1, 100.0), "x": (0.2, 0.9)},
                                            dampening=(0.5, 1.0, 0.8)))):
        rows = []
        for method in methods:
            cfg = t.DynaTrackConfig(tracking_method=method, input_channel="BF", tracking_channel="BF", shift=shift,
                                    segmentation=dict(otsu_sigma=2.0, otsu_component=0),
                                    roi_center=dict(blob_sigma=4.0, background_percentile=50.0, blur_sigma=1.5))
            me = types.SimpleNamespace(_config=cfg, _scale_z=0.17, _scale_yx=0.1133)
            rows.append(t.DynaTrackUpdater._compute_shift(me, ta, tb))
        out[f"compute_shift_{variant}_xyz_um"] = np.array(rows, dtype=np.float64)
    out["limit_shifts_in"] = np.array([[0.3, -5.0, 0.95], [-0.6, 0.05, -0.2], [2.5, 120.0, 0.19]])
    out["limit_shifts_out"] = np.array([t._limit_shifts_zyx(v, {"z": (0.5, 2.0), "y": (0.1, 100.0), "x": (0.2, 0.9)})
                                        for v in out["limit_shifts_in"]])
    np.savez_compressed(GOLD / "ref_dynatrack.npz", **out)
    print("captured ref_dynatrack.npz from", ref)


def main():
    GOLD.mkdir(parents=True, exist_ok=True)
    for name, case in deskew_cases().items():
        np.savez_compressed(GOLD / f"{name}.npz", **case)
    np.savez_compressed(GOLD / "affine_rot2deg.npz", **affine_case())
    np.savez_compressed(GOLD / "rl_5iter.npz", **rl_cases())
    capture_reference_fixtures()
    capture_dynatrack_fixtures()
    for f in sorted(GOLD.glob("*.npz")):
        print(f.name, f.stat().st_size)


if __name__ == "__main__":
    main()
