"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the golden fixtures.

Bars (stated here, as the north-star requires):
* deskew / affine resample: BIT-EXACT vs ``scipy.ndimage.affine_transform(order=1)`` -- the
  kernels evaluate coordinates, weights and the interpolation in fp64 in scipy's own operation
  order, so the float32 results are identical, borders included.
* Richardson-Lucy: float32 FMA stencil vs scipy's float64-accumulating correlate; the error
  compounds multiplicatively over iterations.  Tolerance: ``|gpu - cpu| <= 2e-4 * |cpu| + 1e-4 *
  max|cpu|`` after 20 iterations (5e-5 / 2e-5 after <= 5), separable factorisation included.
"""

import ctypes
import math

import numpy as np
import pytest

from oracle import cpu_ref as o

pytestmark = pytest.mark.gpu


def _t(a, device):
    import torch

    return torch.as_tensor(np.ascontiguousarray(a), device=device)


def _close(gpu, cpu, rtol, atol_rel):
    cpu = np.asarray(cpu, dtype=np.float64)
    gpu = np.asarray(gpu, dtype=np.float64)
    tol = rtol * np.abs(cpu) + atol_rel * np.abs(cpu).max()
    err = np.abs(gpu - cpu)
    worst = np.unravel_index(np.argmax(err - tol), err.shape)
    assert np.all(err <= tol), (
        f"max excess at {worst}: gpu={gpu[worst]!r} cpu={cpu[worst]!r} err={err[worst]:.3e} tol={tol[worst]:.3e}")


# ================================================================ deskew

DESKEW_CASES = ["deskew_nooverhang_avg3", "deskew_overhang_avg1", "deskew_nooverhang_avg5_r0p4"]


@pytest.mark.parametrize("name", DESKEW_CASES)
def test_deskew_matches_golden_bit_exact(device, golden_dir, name):
    from shrimpy_amd.deskew import fast_deskew_zyx

    g = np.load(golden_dir / f"{name}.npz")
    out = fast_deskew_zyx(
        raw_data=_t(g["raw"], device), ls_angle_deg=float(g["ls_angle_deg"]),
        px_to_scan_ratio=float(g["px_to_scan_ratio"]), keep_overhang=bool(g["keep_overhang"]),
        average_n_slices=int(g["average_n_slices"]))
    assert out.device.type == "cuda" and out.dtype.is_floating_point
    np.testing.assert_array_equal(out.cpu().numpy(), g["out"])


@pytest.mark.parametrize("raw_shape,angle,ratio,keep,avg", [
    ((256, 64, 256), 30.0, 0.755, False, 3),   # BASELINE config 1 mapping
    ((200, 70, 130), 30.0, 0.755, True, 3),    # ragged: nothing is a multiple of the tiles
    ((97, 33, 65), 45.0, 1.0, True, 2),        # steepest angle, ratio 1 (z step == 1 row / voxel)
    ((90, 20, 67), 20.0, 1.9, True, 1),        # ratio > 1: narrow-tile path
    ((300, 17, 5), 30.0, 0.3, False, 4),       # tiny X, heavy oversampling, remainder edge-padded
    ((64, 5, 3), 30.0, 0.755, True, 7),        # avg_n larger than the tilt extent
])
def test_deskew_vs_oracle_bit_exact(device, raw_shape, angle, ratio, keep, avg):
    from shrimpy_amd.deskew import fast_deskew_zyx, get_deskewed_data_shape

    rng = np.random.default_rng(hash(raw_shape) % 2**32)
    raw = (rng.random(raw_shape) * 4000 - 500).astype(np.float32)  # negatives too
    ref = o.deskew(raw, angle, ratio, keep, avg)
    out = fast_deskew_zyx(raw_data=_t(raw, device), ls_angle_deg=angle, px_to_scan_ratio=ratio,
                          keep_overhang=keep, average_n_slices=avg)
    shape, _ = get_deskewed_data_shape(raw_shape, angle, ratio, keep, avg)
    assert tuple(out.shape) == shape == ref.shape
    np.testing.assert_array_equal(out.cpu().numpy(), ref)


def test_deskew_general_matrix_falls_back_to_affine_kernels(device):
    """A matrix without the shear structure runs lsr_affine_f32 + lsr_average_slices_f32."""
    from shrimpy_amd.deskew import deskew_with_matrix

    rng = np.random.default_rng(21)
    raw = rng.random((30, 12, 20)).astype(np.float32)
    m, off, pre = o.deskew_geometry(raw.shape, 30.0, 0.755, True)
    m = m.copy()
    m[1, 2] = 0.01  # y_in now depends on X': no longer a pure shear
    ref = o.average_slices(o.affine_apply(raw, m, off, pre), 3)
    out = deskew_with_matrix(_t(raw, device), np.concatenate([m, off[:, None]], 1), pre, 3)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)


def test_deskew_chunks_concatenate_reversed_like_measure_psf(device):
    """``scripts/measure_psf.py:221-249`` on the device path."""
    import torch

    from shrimpy_amd.deskew import fast_deskew_zyx

    rng = np.random.default_rng(23)
    raw = _t(rng.random((80, 24, 64)).astype(np.float32), device)
    kw = dict(ls_angle_deg=30, px_to_scan_ratio=0.755, keep_overhang=True, average_n_slices=3)
    whole = fast_deskew_zyx(raw_data=raw, **kw)
    chunks = [fast_deskew_zyx(raw_data=c.contiguous(), **kw) for c in torch.chunk(raw, 4, dim=-1)]
    assert torch.equal(torch.cat(chunks[::-1], dim=-2), whole)


SWITCH_ORIENTATIONS = ["identity", "flip_z", "flip_y", "flip_x", "transpose_yx", "rot90", "rot180", "rot270",
                       "flip_z+rot90", "transpose_yx+flip_x+flip_z"]


@pytest.mark.parametrize("border", ["constant", "grid-constant"])
@pytest.mark.parametrize("spec", SWITCH_ORIENTATIONS)
def test_deskew_orientation_and_border_switches_vs_oracle_bit_exact(device, spec, border):
    """The two [RECALLED] conventions are Python-level switches: every
    orientation and both border rules against the oracle, float32 and uint16 input, plus the
    ``measure_psf.py:221-249`` chunk identity on the device under that orientation."""
    import torch

    from shrimpy_amd import geometry
    from shrimpy_amd.deskew import fast_deskew_zyx, get_deskewed_data_shape

    rng = np.random.default_rng(31)
    raw = rng.integers(80, 600, (90, 25, 48)).astype(np.float32)
    kw = dict(ls_angle_deg=30.0, px_to_scan_ratio=0.755, keep_overhang=True, average_n_slices=3)
    want = o.deskew(raw, 30.0, 0.755, True, 3, orientation=spec, border=border)
    got = fast_deskew_zyx(raw_data=_t(raw, device), orientation=spec, border=border, **kw)
    assert got.is_contiguous()
    assert tuple(got.shape) == get_deskewed_data_shape(raw.shape, orientation=spec, **kw)[0] == want.shape
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    got16 = fast_deskew_zyx(raw_data=_t(raw.astype(np.uint16), device), orientation=spec, border=border, **kw)
    assert torch.equal(got16, got)
    axis, reverse = geometry.raw_x_chunk_layout(spec)
    parts = [fast_deskew_zyx(raw_data=c.contiguous(), orientation=spec, border=border, **kw)
             for c in torch.chunk(_t(raw, device), 4, dim=-1)]
    assert torch.equal(torch.cat(parts[::-1] if reverse else parts, dim=axis), got)


@pytest.mark.parametrize("border", ["constant", "grid-constant"])
@pytest.mark.parametrize("cval", [7.25, -3.0, "min", None])
def test_deskew_fill_value_vs_oracle_bit_exact(device, cval, border):
    """Round-4 verdict item 6: ``cval`` -- a number or "min" (None is taken as "min": [RECALLED] biahub's
    ``deskew_data(cval=None)`` fills with the stack's minimum) -- against ``scipy.ndimage.affine_transform(cval=...)``
    under both border rules, float32 and uint16 stacks, with and without the overhang, averaging 1 / 3 slices, a ragged
    shape; the minimum is reduced on the device (``lsr_minmax_f32`` / ``lsr_minmax_u16``) and never visits the host."""
    import torch

    from shrimpy_amd.deskew import fast_deskew_zyx

    rng = np.random.default_rng(41)
    for shape, keep, avg in (((90, 25, 48), True, 3), ((300, 40, 70), False, 3), ((257, 33, 65), False, 1)):
        raw = rng.integers(85, 600, shape).astype(np.float32)
        want = o.deskew(raw, 30.0, 0.755, keep, avg, border=border, cval=cval)
        got = fast_deskew_zyx(raw_data=_t(raw, device), ls_angle_deg=30.0, px_to_scan_ratio=0.755, keep_overhang=keep,
                              average_n_slices=avg, border=border, cval=cval)
        assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32)), (shape, keep, avg)
        got16 = fast_deskew_zyx(raw_data=_t(raw.astype(np.uint16), device), ls_angle_deg=30.0, px_to_scan_ratio=0.755,
                                keep_overhang=keep, average_n_slices=avg, border=border, cval=cval)
        assert torch.equal(got16, got)
    assert float(got.min()) >= (85.0 if cval in ("min", None) else min(float(cval), 85.0)) - 1e-3


def test_deskew_fill_value_through_the_pipeline_and_into_the_padded_rl_input(device):
    """``DeskewSettings.cval`` reaches the kernel when the deskew writes the RL kernels' padded input in place."""
    import torch

    from shrimpy_amd.pipeline import VolumeReconstructor
    from shrimpy_amd.settings import DeconvolveSettings, DeskewSettings, ReconstructSettings

    rng = np.random.default_rng(43)
    raw = rng.integers(85, 600, (129, 32, 96)).astype(np.uint16)
    for cval in ("min", 11.0):
        s = ReconstructSettings(deskew=DeskewSettings(pixel_size_um=0.1133, scan_step_um=0.15, ls_angle_deg=30.0,
                                                      keep_overhang=True, average_n_slices=3, cval=cval),
                                deconvolution=DeconvolveSettings(iterations=0))
        got = VolumeReconstructor(raw.shape, s, device)(torch.as_tensor(raw, device=device))
        want = o.deskew(raw.astype(np.float32), 30.0, 0.755, True, 3, cval=cval)
        assert np.array_equal(got.cpu().numpy(), want)
        s1 = s.model_copy(update={"deconvolution": DeconvolveSettings(iterations=1)})
        x1 = VolumeReconstructor(raw.shape, s1, device)(torch.as_tensor(raw, device=device))
        assert tuple(x1.shape) == want.shape and bool(torch.isfinite(x1).all())


@pytest.mark.parametrize("shape,keep,avg", [((90, 25, 48), True, 3), ((300, 40, 70), False, 3), ((64, 16, 130), True, 1),
                                            ((257, 33, 65), False, 2)])
def test_blending_border_runs_the_fused_deskew_kernel(device, shape, keep, avg):
    """``border="grid-constant"`` (scipy ``mode="grid-constant"`` / ``grid_sample(zeros)``: the scan
    continues as zeros, a coordinate within one sample of either end blends) goes through the same
    LDS-transpose kernel as the default rule (``lsr_deskew_border``): equal to the oracle and to the
    general gather kernel + slice averaging bit for bit, for float32 / uint16 input, with the
    flat-field division fused, and when the destination is the RL plan's padded volume."""
    import torch

    from shrimpy_amd import _lib
    from shrimpy_amd.deconvolve import RichardsonLucyPlan
    from shrimpy_amd.deskew import deskew_with_matrix
    from shrimpy_amd.flatfield import flat_field_pattern
    from shrimpy_amd.geometry import deskew_geometry

    rng = np.random.default_rng(41)
    raw = rng.integers(80, 600, shape).astype(np.float32)
    geo = deskew_geometry(shape, 30.0, 0.755, keep, avg, 0.1133)
    want = o.deskew(raw, 30.0, 0.755, keep, avg, border="grid-constant")
    if keep:   # with the overhang kept, output voxels straddle both ends of the scan: the rule matters
        assert not np.array_equal(want, o.deskew(raw, 30.0, 0.755, keep, avg))
    t = _t(raw, device)
    got = deskew_with_matrix(t, geo.matrix_3x4, geo.pre_average_shape, avg, border="grid-constant")
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    # the general kernel + averaging: the pair the fused kernel replaces
    zd, yo, xo = geo.pre_average_shape
    pre = torch.empty((zd, yo, xo), dtype=torch.float32, device=device)
    _lib.call("lsr_affine_f32", t.data_ptr(), *shape, pre.data_ptr(), zd, yo, xo, _lib.matrix12(geo.matrix_3x4),
              ctypes.c_float(0.0), _lib.MODE_GRID_CONSTANT, _lib.stream_ptr(device))
    ref = torch.empty_like(got)
    _lib.call("lsr_average_slices_f32", pre.data_ptr(), zd, yo, xo, ref.data_ptr(), ref.shape[0], avg,
              _lib.stream_ptr(device))
    assert torch.equal(got, ref)
    got16 = deskew_with_matrix(t.to(torch.uint16), geo.matrix_3x4, geo.pre_average_shape, avg, border="grid-constant")
    assert torch.equal(got16, got)
    flat = flat_field_pattern(t)
    fused = deskew_with_matrix(t, geo.matrix_3x4, geo.pre_average_shape, avg, flat_field=flat, border="grid-constant")
    two_step = deskew_with_matrix(flat.apply(t), geo.matrix_3x4, geo.pre_average_shape, avg, border="grid-constant")
    assert torch.equal(fused, two_step)
    psf, _ = o.gaussian_psf((5, 5, 5), (1.2, 1.0, 1.0))
    pl = RichardsonLucyPlan(tuple(got.shape), psf, device)
    if pl.path != "generic":
        padded = pl.new_padded_input()
        deskew_with_matrix(t, geo.matrix_3x4, geo.pre_average_shape, avg, out=padded, border="grid-constant")
        assert torch.equal(padded.view, got)


def test_pipeline_honours_deskew_switches(device):
    """``VolumeReconstructor`` with a rotated, blending deskew in front of RL: output shape, deskew
    bits and the RL result follow the oracle run on the oriented volume."""
    from shrimpy_amd.pipeline import VolumeReconstructor
    from shrimpy_amd.settings import DeconvolveSettings, DeskewSettings, ReconstructSettings

    rng = np.random.default_rng(33)
    raw = rng.integers(80, 600, (96, 24, 40)).astype(np.uint16)
    d = DeskewSettings(pixel_size_um=0.1133, scan_step_um=0.15, ls_angle_deg=30.0, keep_overhang=True,
                       average_n_slices=3, orientation="rot90", border="grid-constant")
    rec = VolumeReconstructor(raw.shape, ReconstructSettings(deskew=d), device)
    want = o.deskew(raw.astype(np.float32), 30.0, 0.755, True, 3, orientation="rot90", border="grid-constant")
    assert rec.output_shape == want.shape
    np.testing.assert_array_equal(rec(raw).cpu().numpy(), want)
    rec = VolumeReconstructor(raw.shape, ReconstructSettings(
        deskew=d, deconvolution=DeconvolveSettings(iterations=5)), device)
    psf, _ = o.gaussian_psf((9, 7, 7), (2.0, 1.2, 1.2))
    _close(rec(raw).cpu().numpy(), o.richardson_lucy(want, psf, iterations=5), 5e-5, 2e-5)


def test_deskew_errors(device):
    import torch

    from shrimpy_amd.deskew import fast_deskew_zyx

    with pytest.raises(ValueError, match="empty"):
        fast_deskew_zyx(raw_data=torch.zeros((8, 64, 4), device=device), ls_angle_deg=30,
                        px_to_scan_ratio=0.755, keep_overhang=False)
    with pytest.raises(ValueError):
        fast_deskew_zyx(raw_data=torch.zeros((8, 8), device=device), ls_angle_deg=30,
                        px_to_scan_ratio=0.755, keep_overhang=True)
    # uint16 camera frames are converted like the reference does (torch.as_tensor(..., float32))
    raw = torch.arange(40 * 6 * 8, device=device).reshape(40, 6, 8).to(torch.int32)
    out = fast_deskew_zyx(raw_data=raw, ls_angle_deg=30, px_to_scan_ratio=0.755, keep_overhang=True)
    ref = o.deskew(raw.cpu().numpy().astype(np.float32), 30, 0.755, True, 1)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)


# ================================================================ affine


def test_affine_matches_golden_bit_exact(device, golden_dir):
    from shrimpy_amd.register import apply_affine_transform_zyx

    g = np.load(golden_dir / "affine_rot2deg.npz")
    vol = _t(g["vol"], device)
    out = apply_affine_transform_zyx(vol, g["matrix"])
    np.testing.assert_array_equal(out.cpu().numpy(), g["out_constant"])
    out = apply_affine_transform_zyx(vol, g["matrix"], tuple(g["grid_shape"]), mode="grid-constant",
                                     cval=float(g["grid_cval"]))
    np.testing.assert_array_equal(out.cpu().numpy(), g["out_grid"])


def _config3_matrix():
    """rotation 2 deg about Z, scale (1, .98, 1.02), translation (3.5,-12.25,20.75)."""
    th = np.deg2rad(2.0)
    rot = np.array([[1, 0, 0], [0, np.cos(th), -np.sin(th)], [0, np.sin(th), np.cos(th)]])
    m = np.eye(4)
    m[:3, :3] = rot @ np.diag([1.0, 0.98, 1.02])
    m[:3, 3] = [3.5, -12.25, 20.75]
    return m


@pytest.mark.parametrize("mode", ["constant", "grid-constant"])
@pytest.mark.parametrize("shape,out_shape", [((20, 96, 130), None), ((9, 33, 70), (12, 40, 64))])
def test_affine_vs_oracle_bit_exact(device, mode, shape, out_shape):
    from shrimpy_amd.register import apply_affine_transform_zyx

    rng = np.random.default_rng(31)
    vol = (rng.random(shape) * 1000 - 100).astype(np.float32)
    m = _config3_matrix()
    oshape = out_shape or shape
    ref = o.affine_apply_4x4(vol, m, oshape, cval=-3.0, mode=mode)
    out = apply_affine_transform_zyx(_t(vol, device), m, oshape, mode=mode, cval=-3.0)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)


def _planar_matrix(theta_deg, scale_zyx, shift_zyx, shear=0.0):
    th = np.deg2rad(theta_deg)
    m = np.eye(4)
    m[1:3, 1:3] = np.array([[np.cos(th), -np.sin(th)], [np.sin(th) + shear, np.cos(th)]]) @ np.diag(scale_zyx[1:])
    m[0, 0] = scale_zyx[0]
    m[:3, 3] = shift_zyx
    return m


@pytest.mark.parametrize("case", [
    dict(shape=(20, 96, 132), theta=2.0, scale=(1.0, 0.98, 1.02), shift=(3.5, -12.25, 20.75)),
    dict(shape=(9, 40, 72), out=(12, 40, 64), theta=-7.0, scale=(1.0, 1.0, 1.0), shift=(0.0, 3.0, -2.0)),
    dict(shape=(17, 70, 260), theta=0.0, scale=(-1.0, 1.0, 1.0), shift=(16.0, 0.0, 0.0)),        # z flip, identity plane
    dict(shape=(17, 70, 260), theta=1.0, scale=(0.5, 1.01, 0.99), shift=(0.25, 0.5, 0.5)),        # z upsampling
    dict(shape=(30, 33, 68), out=(24, 50, 80), theta=3.0, scale=(1.3, 0.9, 1.1), shift=(-2.5, -6.0, 4.0), shear=0.05),
    dict(shape=(6, 64, 128), theta=0.0, scale=(0.0, 1.0, 1.0), shift=(2.0, 0.0, 0.0)),            # every plane from z = 2
    dict(shape=(5, 20, 16), out=(5, 45, 300), theta=30.0, scale=(1.0, 0.4, 0.05), shift=(0.0, 0.0, 0.0)),
    dict(shape=(4, 300, 2000), out=(4, 20, 140), theta=5.0, scale=(1.0, 9.0, 14.0), shift=(0.0, 3.0, 1.0)),  # box > LDS
    dict(shape=(8, 48, 96), theta=12.0, scale=(1.0, 1.0, 1.0), shift=(0.0, 500.0, 0.0)),          # everything outside
    # maps that decimate in the plane: the smaller tiles of round 4 (32 x 64, 16 x 128, 16 x 64) keep them off the gather kernel
    dict(shape=(6, 200, 520), out=(6, 100, 260), theta=0.0, scale=(1.0, 2.0, 2.0), shift=(0.0, 0.25, 0.5)),
    dict(shape=(5, 260, 532), out=(7, 70, 200), theta=3.0, scale=(0.9, 3.0, 2.5), shift=(0.5, 4.0, 3.0), shear=0.03),
    dict(shape=(4, 300, 700), out=(4, 61, 129), theta=-2.0, scale=(1.0, 4.5, 5.0), shift=(0.0, 2.0, 1.0)),
])
@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("mode", ["constant", "grid-constant"])
def test_affine_planar_kernel_vs_oracle(device, case, exact, mode):
    """z-decoupled maps run affine_planar.hip (LDS-staged source boxes, z march) under either border rule:
    bit-identical to scipy in exact mode, ~1e-6 with f32 interpolation, same border decisions."""
    from shrimpy_amd import _lib
    from shrimpy_amd.geometry import as_matrix_3x4
    from shrimpy_amd.register import apply_affine_transform_zyx

    code = _lib.MODE_CONSTANT if mode == "constant" else _lib.MODE_GRID_CONSTANT
    planar = _lib.call_value("lsr_affine_kernel_choice", case["shape"][1], case["shape"][2],
                             _lib.matrix12(as_matrix_3x4(_planar_matrix(case["theta"], case["scale"], case["shift"],
                                                                        case.get("shear", 0.0)))), code)
    # (beyond ~4x decimation the source box of even a 16 x 64 tile exceeds the ring's LDS: the gather kernel)
    assert planar == (0 if case["scale"][2] > 4 else 1)
    rng = np.random.default_rng(hash(str(case)) % 2**32)
    vol = (rng.random(case["shape"]) * 1000 - 100).astype(np.float32)
    m = _planar_matrix(case["theta"], case["scale"], case["shift"], case.get("shear", 0.0))
    oshape = case.get("out", case["shape"])
    ref = o.affine_apply_4x4(vol, m, oshape, cval=-3.0, mode=mode)
    out = apply_affine_transform_zyx(_t(vol, device), m, oshape, cval=-3.0, exact=exact, mode=mode).cpu().numpy()
    if exact:
        np.testing.assert_array_equal(out, ref)
    elif mode == "constant":
        assert np.array_equal(out == -3.0, ref == -3.0)
        np.testing.assert_allclose(out, ref, rtol=2e-5, atol=2e-3)
    else:   # blended borders: no exact cval pattern to compare, the values carry the decision
        np.testing.assert_allclose(out, ref, rtol=2e-5, atol=2e-3)


def _rotation(axis, deg):
    th = np.deg2rad(deg)
    c, s = np.cos(th), np.sin(th)
    i, j = [(1, 2), (0, 2), (0, 1)][axis]       # rotate in the plane of the two other axes
    r = np.eye(3)
    r[i, i], r[i, j], r[j, i], r[j, j] = c, -s, s, c
    return r


def _tilted_matrix(rots, scale=(1.0, 1.0, 1.0), shift=(0.0, 0.0, 0.0), centre=None):
    """Rotations about (axis, degrees) in sequence, then a diagonal scale; about ``centre`` if given."""
    lin = np.eye(3)
    for axis, deg in rots:
        lin = lin @ _rotation(axis, deg)
    lin = lin @ np.diag(scale)
    m = np.eye(4)
    m[:3, :3] = lin
    m[:3, 3] = shift
    if centre is not None:
        c = np.asarray(centre, dtype=np.float64)
        m[:3, 3] = c - lin @ c + np.asarray(shift)
    return m


BOX_CASES = [
    # a tilted registration: light-sheet volume rotated 3 deg about Y (couples z and x), 2 deg in the plane
    dict(shape=(24, 96, 132), m=_tilted_matrix([(1, 3.0), (0, 2.0)], (1.0, 0.98, 1.02), (1.5, -4.25, 6.75))),
    # about the volume centre, all three axes, ragged output larger than the input
    dict(shape=(19, 45, 76), out=(23, 50, 91), m=_tilted_matrix([(0, 4.0), (1, -6.0), (2, 5.0)], centre=(9, 22, 38))),
    # z depends on (y, x) only through a shear; the plane does not depend on z
    dict(shape=(16, 40, 64), m=np.array([[1, 0.11, -0.07, 2.0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])),
    # the plane depends on z; z decoupled
    dict(shape=(16, 40, 64), m=np.array([[1, 0, 0, 0.5], [0.3, 1, 0, -1.0], [-0.2, 0, 1, 2.0], [0, 0, 0, 1.0]])),
    # flips on every axis plus a tilt
    dict(shape=(12, 33, 68), m=_tilted_matrix([(1, 2.0)], (-1.0, -1.0, -1.0), (11.0, 32.0, 67.0))),
    # downsampling 1.3x with a tilt: a large (but fitting) source box, one workgroup per CU
    dict(shape=(40, 90, 200), out=(28, 66, 150), m=_tilted_matrix([(1, 1.5), (2, 1.0)], (1.3, 1.3, 1.3))),
    # upsampling 3x: tiny source boxes, many output blocks per source voxel
    dict(shape=(6, 10, 24), out=(17, 28, 70), m=_tilted_matrix([(1, 8.0)], (0.33, 0.33, 0.33))),
    # 30 degrees about X: steep coupling of z and y
    dict(shape=(30, 30, 64), m=_tilted_matrix([(2, 30.0)], centre=(15, 15, 32))),
    # pushed far away: every block is blind
    dict(shape=(8, 24, 64), m=_tilted_matrix([(1, 3.0)], shift=(0.0, 0.0, 5000.0))),
    # one plane / one row outputs
    dict(shape=(9, 17, 32), out=(1, 1, 130), m=_tilted_matrix([(1, 3.0), (2, 2.0)], (1.0, 1.0, 0.24), (3.0, 5.0, 0.0))),
]


@pytest.mark.parametrize("case", BOX_CASES)
@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("mode", ["constant", "grid-constant"])
def test_affine_box_kernel_vs_oracle(device, case, exact, mode):
    """Maps that couple z with the plane (and any other map whose per-block source box fits in LDS)
    run affine_box.hip under either border rule: bit-identical to scipy in exact mode, within 2e-5 of the
    data range with f32 interpolation and with the same in / out-of-range decisions."""
    from shrimpy_amd import _lib
    from shrimpy_amd.geometry import as_matrix_3x4
    from shrimpy_amd.register import apply_affine_transform_zyx

    m, shape = case["m"], case["shape"]
    path = _lib.call_value("lsr_affine_path", shape[0], shape[1], shape[2], _lib.matrix12(as_matrix_3x4(m)),
                           _lib.MODE_CONSTANT if mode == "constant" else _lib.MODE_GRID_CONSTANT)
    assert path == 2
    rng = np.random.default_rng(len(str(case)))
    vol = (rng.random(shape) * 1000 - 100).astype(np.float32)
    oshape = case.get("out", shape)
    ref = o.affine_apply_4x4(vol, m, oshape, cval=-3.0, mode=mode)
    out = apply_affine_transform_zyx(_t(vol, device), m, oshape, cval=-3.0, exact=exact, mode=mode).cpu().numpy()
    if exact:
        np.testing.assert_array_equal(out, ref)
    else:
        if mode == "constant":
            assert np.array_equal(out == -3.0, ref == -3.0)
        assert np.abs(out - ref).max() <= 2e-5 * 1100


PITCHED_CASES = [
    dict(shape=(24, 96, 130), m="config3", path=1),                       # planar, width 130 = 4 * 32 + 2
    dict(shape=(20, 70, 133), out=(22, 75, 140), m="config3", path=1),
    dict(shape=(24, 96, 134), m=_tilted_matrix([(1, 3.0), (0, 2.0)], (1.0, 0.98, 1.02), (1.5, -4.25, 6.75)), path=2),
    dict(shape=(19, 45, 77), out=(23, 50, 91), m=_tilted_matrix([(0, 4.0), (1, -6.0), (2, 5.0)], centre=(9, 22, 38)), path=2),
    dict(shape=(12, 33, 67), m=_tilted_matrix([(1, 2.0)], (-1.0, -1.0, -1.0), (11.0, 32.0, 66.0)), path=2),   # flips: taps at both ends of a row
    dict(shape=(16, 40, 9), m=_tilted_matrix([(1, 2.0)]), path=2),        # narrower than three chunks
]


@pytest.mark.parametrize("case", PITCHED_CASES)
@pytest.mark.parametrize("exact", [True, False])
def test_affine_over_padded_rows_keeps_the_lds_kernels(device, case, exact):
    """A moving volume whose width is not a multiple of 4 (three deskewed volumes in four) goes to the
    gather kernel through ``lsr_affine_f32``; with its rows padded to a multiple of 4
    (``register.PitchedVolume``, ``lsr_affine_pitched_f32``) the planar / box kernels take it: same
    bits as the oracle, whatever finite values sit in the padding."""
    import torch

    from shrimpy_amd import _lib
    from shrimpy_amd.geometry import as_matrix_3x4
    from shrimpy_amd.register import PitchedVolume, apply_affine_transform_zyx

    shape = case["shape"]
    m = _config3_matrix() if isinstance(case["m"], str) else case["m"]
    m12 = _lib.matrix12(as_matrix_3x4(m))
    assert _lib.call_value("lsr_affine_path", *shape, m12, _lib.MODE_CONSTANT) == 0
    rng = np.random.default_rng(len(str(case)))
    vol = (rng.random(shape) * 1000 - 100).astype(np.float32)
    oshape = case.get("out", shape)
    ref = o.affine_apply_4x4(vol, m, oshape, cval=-3.0, mode="constant")
    src = PitchedVolume.copy_of(_t(vol, device))
    assert src.pitch % 4 == 0 and src.pitch - shape[2] in (1, 2, 3) and tuple(src.shape) == shape
    assert _lib.call_value("lsr_affine_path_pitched", *shape, src.pitch, src.plane, m12, _lib.MODE_CONSTANT) == case["path"]

    def check(out):
        out = out.cpu().numpy()
        if exact:
            np.testing.assert_array_equal(out, ref)
        else:
            assert np.array_equal(out == -3.0, ref == -3.0)
            assert np.abs(out - ref).max() <= 2e-5 * 1100

    check(apply_affine_transform_zyx(src, m, oshape, cval=-3.0, exact=exact))
    src.full[:, :, shape[2]:] = 4321.0              # the padding only ever meets weight 0
    check(apply_affine_transform_zyx(src, m, oshape, cval=-3.0, exact=exact))
    # a dense tensor of that width keeps the gather kernel (a padded copy costs more than it saves): same bits
    check(apply_affine_transform_zyx(_t(vol, device), m, oshape, cval=-3.0, exact=exact))


def test_affine_pitched_argument_errors(device):
    import torch

    from shrimpy_amd import _lib
    from shrimpy_amd.geometry import as_matrix_3x4

    a = torch.zeros((4, 8, 12), device=device)
    b = torch.zeros((4, 8, 10), device=device)
    m12 = _lib.matrix12(as_matrix_3x4(np.eye(4)))
    for pitch, plane in ((9, 96), (12, 80)):
        with pytest.raises(_lib.LsrError, match="source strides"):
            _lib.call("lsr_affine_pitched_f32", a.data_ptr(), 4, 8, 10, pitch, plane, b.data_ptr(), 4, 8, 10, 10, 80, m12,
                      ctypes.c_float(0.0), _lib.MODE_CONSTANT, _lib.stream_ptr(device))
    for pitch, plane in ((9, 80), (10, 79)):
        with pytest.raises(_lib.LsrError, match="output strides"):
            _lib.call("lsr_affine_pitched_f32", a.data_ptr(), 4, 8, 10, 12, 96, b.data_ptr(), 4, 8, 10, pitch, plane, m12,
                      ctypes.c_float(0.0), _lib.MODE_CONSTANT, _lib.stream_ptr(device))
    # strides the LDS kernels cannot take (pitch 11: rows off 16-byte boundaries) still give the right answer
    c = torch.arange(4 * 8 * 11, dtype=torch.float32, device=device).reshape(4, 8, 11)
    _lib.call("lsr_affine_pitched_f32", c.data_ptr(), 4, 8, 10, 11, 88, b.data_ptr(), 4, 8, 10, 10, 80, m12,
              ctypes.c_float(0.0), _lib.MODE_CONSTANT, _lib.stream_ptr(device))
    assert torch.equal(b, c[:, :, :10])


@pytest.mark.parametrize("mode,cval", [("constant", 0.0), ("grid-constant", 0.0), ("grid-constant", 7.5)])
def test_pipeline_deskews_into_padded_rows_when_a_registration_follows(device, mode, cval):
    """deskew -> register on a stack whose deskewed width is not a multiple of 4: the deskew writes
    zero-padded rows and the registration reads them through the LDS-staged kernel (either border rule:
    padding columns are never a tap's value); result = oracle."""
    from shrimpy_amd.pipeline import VolumeReconstructor
    from shrimpy_amd.settings import DeskewSettings, ReconstructSettings, RegisterSettings

    rng = np.random.default_rng(35)
    raw = rng.integers(80, 600, (98, 24, 40)).astype(np.uint16)        # deskews to (8, 40, 151)
    d = DeskewSettings(pixel_size_um=0.1133, scan_step_um=0.15, ls_angle_deg=30.0, keep_overhang=True, average_n_slices=3)
    desk = o.deskew(raw.astype(np.float32), 30.0, 0.755, True, 3)
    assert desk.shape[2] % 4 != 0
    m = _tilted_matrix([(1, 2.0), (0, 1.0)], (1.0, 0.99, 1.01), (0.5, -1.25, 2.75))
    reg = RegisterSettings(affine_transform_zyx=m.tolist(), mode=mode, cval=cval)
    rec = VolumeReconstructor(raw.shape, ReconstructSettings(deskew=d, registration=reg), device)
    got = rec(raw)
    assert rec._pitched is not None and rec._pitched.pitch == (desk.shape[2] + 3) // 4 * 4
    np.testing.assert_array_equal(got.cpu().numpy(), o.affine_apply_4x4(desk, m, desk.shape, cval=cval, mode=mode))
    np.testing.assert_array_equal(rec(raw).cpu().numpy(), got.cpu().numpy())     # the padded target is reused


def test_affine_path_selection(device):
    """planar (1) for z-decoupled maps, box (2) for the rest that fit, gather (0) otherwise; every
    path gives the oracle's bits."""
    from shrimpy_amd import _lib
    from shrimpy_amd.geometry import as_matrix_3x4
    from shrimpy_amd.register import apply_affine_transform_zyx

    def path(shape, m, mode=_lib.MODE_CONSTANT):
        return _lib.call_value("lsr_affine_path", shape[0], shape[1], shape[2], _lib.matrix12(as_matrix_3x4(m)), mode)

    tilt = _tilted_matrix([(1, 3.0)])
    assert path((20, 96, 132), _config3_matrix()) == 1
    assert path((20, 96, 132), tilt) == 2
    assert path((20, 96, 130), tilt) == 0                                 # Xi not a multiple of 4
    assert path((20, 96, 132), tilt, _lib.MODE_GRID_CONSTANT) == 2        # the blending border rule too
    assert path((20, 96, 132), _tilted_matrix([(1, 3.0)], (9.0, 9.0, 9.0))) == 0   # source box beyond LDS
    rng = np.random.default_rng(77)
    for shape in ((20, 96, 132), (20, 96, 130)):
        vol = (rng.random(shape) * 1000).astype(np.float32)
        out = apply_affine_transform_zyx(_t(vol, device), tilt).cpu().numpy()
        np.testing.assert_array_equal(out, o.affine_apply_4x4(vol, tilt, shape))


@pytest.mark.parametrize("mode", ["constant", "grid-constant"])
def test_affine_f32_interpolation_mode_is_close_and_keeps_the_border(device, mode):
    """``exact=False``: f32 interpolation; tolerance 2e-5 of the data range, and the
    in/out-of-range decisions (exact zeros of mode="constant") are unchanged."""
    from shrimpy_amd.register import apply_affine_transform_zyx

    rng = np.random.default_rng(35)
    vol = (rng.random((20, 96, 130)) * 1000).astype(np.float32)
    m = _config3_matrix()
    ref = o.affine_apply_4x4(vol, m, vol.shape, cval=0.0, mode=mode)
    out = apply_affine_transform_zyx(_t(vol, device), m, mode=mode, exact=False).cpu().numpy()
    assert np.abs(out - ref).max() <= 2e-5 * 1000
    if mode == "constant":
        np.testing.assert_array_equal(out == 0, ref == 0)


def test_affine_border_rule_exact_on_grid_points(device):
    """Coordinates that land exactly on 0 and n-1 are inside; -1e-9 is outside."""
    from shrimpy_amd.register import affine_transform

    vol = np.arange(5 * 6 * 7, dtype=np.float32).reshape(5, 6, 7) + 1
    for off in ([0.0, 0.0, 0.0], [-1e-9, 0.0, 0.0], [1.0, -2.0, 3.0], [0.0, 1e-12, 0.0]):
        ref = o.affine_apply(vol, np.eye(3), np.array(off), vol.shape)
        out = affine_transform(_t(vol, device), np.eye(3), offset=off)
        np.testing.assert_array_equal(out.cpu().numpy(), ref)
    # anisotropic scaling via a diagonal, output larger than input
    ref = o.affine_apply(vol, np.diag([0.5, 1.25, 0.75]), np.zeros(3), (9, 5, 9))
    out = affine_transform(_t(vol, device), [0.5, 1.25, 0.75], output_shape=(9, 5, 9))
    np.testing.assert_array_equal(out.cpu().numpy(), ref)


def test_average_n_slices_vs_oracle(device):
    from shrimpy_amd.deskew import average_n_slices

    rng = np.random.default_rng(33)
    d = rng.random((11, 9, 70)).astype(np.float32)
    for n in (1, 2, 3, 4, 11, 16):
        out = average_n_slices(_t(d, device), n)
        np.testing.assert_array_equal(out.cpu().numpy(), o.average_slices(d, n))


# ================================================================ correlation + Richardson-Lucy


@pytest.mark.parametrize("pshape", [(1, 1, 1), (3, 3, 3), (9, 7, 7), (5, 3, 9), (15, 15, 15), (3, 1, 5), (11, 5, 13), (13, 9, 3)])
@pytest.mark.parametrize("vshape", [(20, 40, 70), (3, 5, 4)])
def test_correlate_sep_and_dense_vs_scipy(device, pshape, vshape):
    from scipy import ndimage

    from shrimpy_amd.deconvolve import correlate3d

    rng = np.random.default_rng(41)
    vol = rng.random(vshape).astype(np.float32)
    ks = [rng.random(n).astype(np.float32) + 0.1 for n in pshape]
    w = (ks[0][:, None, None] * ks[1][None, :, None] * ks[2][None, None, :]).astype(np.float32)
    ref = ndimage.correlate(vol, w, mode="constant", cval=0.0)
    dense = correlate3d(_t(vol, device), w).cpu().numpy()            # tuned where pz<=11, pyx<=9
    _close(dense, ref, 2e-5, 2e-6)
    generic = correlate3d(_t(vol, device), w, tuned=False).cpu().numpy()
    _close(generic, ref, 2e-5, 2e-6)
    sep = correlate3d(_t(vol, device), weight_factors=ks).cpu().numpy()
    _close(sep, ref, 2e-5, 2e-6)


def test_correlate_long_z_runs_several_z_chunks(device):
    """Few tiles and a long z axis: the launch splits z into chunks, each re-reading PZ-1 halo
    planes; results must not depend on the split."""
    from scipy import ndimage

    from shrimpy_amd.deconvolve import correlate3d, richardson_lucy

    rng = np.random.default_rng(45)
    vol = (rng.random((230, 40, 70)) + 0.5).astype(np.float32)
    ks = [rng.random(n).astype(np.float32) + 0.1 for n in (9, 7, 7)]
    ks = [k / k.sum() for k in ks]
    ref = vol
    for axis, k in enumerate(ks):
        ref = ndimage.correlate1d(ref, k, axis=axis, mode="constant", cval=0.0)
    _close(correlate3d(_t(vol, device), weight_factors=ks).cpu().numpy(), ref, 2e-5, 2e-6)
    x = richardson_lucy(_t(vol, device), psf_factors=ks, iterations=3).cpu().numpy()
    _close(x, o.richardson_lucy_separable(vol, ks, 3), 5e-5, 2e-5)


def test_correlate_is_not_convolve(device):
    """Asymmetric kernel: a flipped implementation fails this."""
    from scipy import ndimage

    from shrimpy_amd.deconvolve import correlate3d

    vol = np.zeros((7, 9, 11), np.float32)
    vol[3, 4, 5] = 1
    w = np.arange(27, dtype=np.float32).reshape(3, 3, 3)
    out = correlate3d(_t(vol, device), w).cpu().numpy()
    np.testing.assert_allclose(out, ndimage.correlate(vol, w, mode="constant"), atol=1e-6)
    assert not np.allclose(out, ndimage.convolve(vol, w, mode="constant"))


def test_rl_matches_golden(device, golden_dir):
    from shrimpy_amd.deconvolve import richardson_lucy

    g = np.load(golden_dir / "rl_5iter.npz")
    y = _t(g["y"], device)
    x = richardson_lucy(y, g["psf_sep"], iterations=5)
    _close(x.cpu().numpy(), g["x_sep_5"], 5e-5, 2e-5)
    x = richardson_lucy(y, psf_factors=(g["kz"], g["ky"], g["kx"]), iterations=5)
    _close(x.cpu().numpy(), g["x_sep_5"], 5e-5, 2e-5)
    x = richardson_lucy(y, g["psf_rot"], iterations=5)
    _close(x.cpu().numpy(), g["x_rot_5"], 5e-5, 2e-5)
    x = richardson_lucy(y, g["psf_rot"], iterations=1)
    _close(x.cpu().numpy(), g["x_rot_1"], 2e-5, 5e-6)


@pytest.mark.parametrize("separable", ["auto", "never"])
def test_rl_20_iterations_vs_oracle(device, separable):
    from shrimpy_amd.deconvolve import RichardsonLucyPlan

    psf, _ = o.gaussian_psf((9, 7, 7), (2.0, 1.2, 1.2))
    y = o.bead_scene((24, 48, 80), seed=2007, psf=psf, density=5e-4)
    ref = o.richardson_lucy(y, psf, iterations=20)
    plan = RichardsonLucyPlan(y.shape, psf, device, separable=separable)
    assert plan.separable == (separable == "auto")
    x = plan(_t(y, device), iterations=20)
    _close(x.cpu().numpy(), ref, 2e-4, 1e-4)
    assert float(x.min()) >= 0


@pytest.mark.parametrize("pshape", [(9, 7, 7), (3, 3, 3), (5, 3, 9), (1, 1, 1), (9, 9, 9), (7, 5, 5), (9, 1, 7),
                                    (11, 7, 13), (13, 15, 15), (15, 9, 3), (9, 11, 5), (3, 15, 11), (11, 11, 11)])
@pytest.mark.parametrize("vshape", [(20, 40, 70), (3, 5, 4), (37, 70, 300), (11, 33, 129)])
def test_rl_fused_equals_two_launch_bit_exact(device, pshape, vshape):
    """One launch per iteration (rl_fused_sep.hip) keeps the per-voxel arithmetic of the
    ratio / update launches: the volumes are equal bit for bit, for x0 = y and for a given x0,
    with a dense y (copied into a padded volume) and with y already padded, for odd and even
    iteration counts (the working volumes ping-pong)."""
    import torch

    from shrimpy_amd.deconvolve import RichardsonLucyPlan

    rng = np.random.default_rng(sum(pshape) * 100 + sum(vshape))
    factors = [np.abs(rng.normal(1.0, 0.4, n)).astype(np.float32) + 0.05 for n in pshape]
    factors = [f / f.sum() for f in factors]           # asymmetric taps: flipped != unflipped
    y = _t((rng.random(vshape) * 80 + 1).astype(np.float32), device)
    x0 = _t((rng.random(vshape) * 40 + 1).astype(np.float32), device)
    # ("always": the default leaves wide in-plane extents with a short z extent to the pair, deconvolve.fused_pays)
    fused = RichardsonLucyPlan(vshape, None, device, psf_factors=factors, fused="always")
    plain = RichardsonLucyPlan(vshape, None, device, psf_factors=factors, fused="never")
    assert fused.fused and not plain.fused
    for iters in (1, 2, 5):
        assert torch.equal(fused(y, iterations=iters), plain(y, iterations=iters))
    assert torch.equal(fused(y, iterations=3, x0=x0), plain(y, iterations=3, x0=x0))
    ypad = fused.new_padded_input()
    ypad.view.copy_(y)
    assert torch.equal(fused(ypad, iterations=4), plain(y, iterations=4))
    assert torch.equal(ypad.view, y)                     # y is read, never written


def test_rl_fused_equals_two_launch_on_random_shapes(device):
    """Seeded sweep over awkward sizes: volumes thinner than the PSF, one-plane volumes, extents
    one off a tile multiple, every tile-size class of the fused kernel."""
    import torch

    from shrimpy_amd.deconvolve import RichardsonLucyPlan

    rng = np.random.default_rng(20260101)
    odd = np.array([1, 3, 5, 7, 9, 11, 13, 15])
    for case in range(24):
        pshape = tuple(int(v) for v in rng.choice(odd, 3))
        if pshape[0] >= 15 and max(pshape[1:]) >= 11:
            pshape = (13,) + pshape[1:]
        vshape = (int(rng.integers(1, 40)), int(rng.choice([1, 2, 15, 16, 17, 31, 33, 47, 49, 64, 65])),
                  int(rng.choice([1, 3, 63, 64, 65, 127, 128, 129, 200, 257])))
        factors = [np.abs(rng.normal(1.0, 0.4, n)).astype(np.float32) + 0.05 for n in pshape]
        factors = [f / f.sum() for f in factors]
        y = _t((rng.random(vshape) * 80 + 1).astype(np.float32), device)
        iters = int(rng.integers(1, 4))
        a = RichardsonLucyPlan(vshape, None, device, psf_factors=factors, fused="always")(y, iterations=iters)
        b = RichardsonLucyPlan(vshape, None, device, psf_factors=factors, fused="never")(y, iterations=iters)
        assert torch.equal(a, b), (case, pshape, vshape, iters)


def test_rl_fused_covers_every_separable_psf(device):
    """Every odd tap count up to 15 per axis has a fused specialisation (smaller tiles for the
    larger PSFs); the two-launch kernels stay as the cross-check (fused="never")."""
    from shrimpy_amd import _lib
    from shrimpy_amd.deconvolve import RichardsonLucyPlan

    for taps in ((9, 7, 7), (11, 7, 7), (9, 11, 3), (13, 15, 15), (15, 9, 9), (1, 1, 1)):
        assert _lib.call_value("lsr_rl_sep_fused_supported", *taps) == 1
    assert _lib.call_value("lsr_rl_sep_fused_supported", 15, 15, 15) == 0   # would spill: two-launch path
    assert _lib.call_value("lsr_rl_sep_fused_supported", 17, 3, 3) == 0
    assert _lib.call_value("lsr_rl_sep_fused_supported", 4, 3, 3) == 0
    from shrimpy_amd.deconvolve import fused_pays

    # the default takes the one-launch form where it is the faster one (measured: profiles/r03_rl_psf_sweep.jsonl)
    assert fused_pays(9, 7, 7) and fused_pays(13, 15, 15) and fused_pays(3, 11, 11)
    assert not fused_pays(5, 13, 13) and not fused_pays(9, 15, 3)
    wide = RichardsonLucyPlan((16, 30, 50), None, device, psf_factors=o.gaussian_psf((5, 13, 13), (1.0, 2.5, 2.5))[1])
    assert wide.separable and not wide.fused and wide.path == "separable"
    psf, factors = o.gaussian_psf((11, 7, 13), (2.0, 1.2, 2.5))
    plan = RichardsonLucyPlan((16, 30, 50), None, device, psf_factors=factors)
    assert plan.separable and plan.fused
    y = o.bead_scene((16, 30, 50), seed=5, psf=psf, density=1e-3)
    _close(plan(_t(y, device), iterations=3).cpu().numpy(), o.richardson_lucy(y, psf, 3), 5e-5, 2e-5)


def test_rl_dense_rotated_psf_20_iterations(device):
    from shrimpy_amd.deconvolve import richardson_lucy

    psf = o.rotated_psf((9, 7, 7), (2.0, 1.2, 1.2), 30.0)
    y = o.bead_scene((20, 40, 72), seed=2009, psf=psf, density=5e-4)
    ref = o.richardson_lucy(y, psf, iterations=20)
    x = richardson_lucy(_t(y, device), psf, iterations=20)
    _close(x.cpu().numpy(), ref, 2e-4, 1e-4)


@pytest.mark.parametrize("pshape", [(11, 9, 9), (11, 7, 9), (9, 9, 9), (11, 9, 3)])
def test_rl_tuned_dense_kernel_at_its_largest_extents_vs_oracle(device, pshape):
    """The tuned dense kernel's biggest instances (11 z taps, 9 in plane: 891 FMAs per voxel).  Round 4's static check
    (tools/asm_hazards.py) found <11, 9, UPDATE> compiled with its pipeline lambda out of line -- staging registers in
    scratch memory, stored there before the loads into them had landed -- and no test had ever run that instance."""
    from shrimpy_amd.deconvolve import RichardsonLucyPlan

    psf = o.rotated_psf(pshape, (2.3, 1.6, 1.6), 30.0)
    psf[0, 0, 0] += 0.003            # neither rank-1 nor ky (x) kzx
    psf /= psf.sum()
    vshape = (14, 40, 70)
    y = o.bead_scene(vshape, seed=sum(pshape), psf=o.gaussian_psf((5, 5, 5), (1.2, 1.0, 1.0))[0], density=2e-3)
    plan = RichardsonLucyPlan(vshape, psf, device)
    assert plan.path == "dense"
    x = plan(_t(y, device), iterations=3)
    _close(x.cpu().numpy(), o.richardson_lucy(y, psf, 3), 5e-5, 2e-5)
    again = plan(_t(y, device), iterations=3)
    assert np.array_equal(x.cpu().numpy(), again.cpu().numpy())


@pytest.mark.parametrize("pshape,sigma", [((31, 7, 7), (6.0, 1.2, 1.2)), ((21, 15, 15), (4.0, 2.5, 2.5)), ((17, 3, 5), (3.0, 0.8, 1.1))])
def test_rl_separable_psf_with_a_long_z_factor(device, pshape, sigma):
    """Separable PSFs with 17 .. 31 z taps: every correlation = in-plane launch (one z tap) + the z march of
    csrc/correlate_z.hip.  Against the oracle within the RL bar, scalars included; thin volumes (fewer planes than taps),
    a padded y, an explicit x0, tol."""
    import torch

    from shrimpy_amd.deconvolve import RichardsonLucyPlan

    psf, factors = o.gaussian_psf(pshape, sigma)
    vshape = (40, 37, 131)
    y = o.bead_scene(vshape, seed=sum(pshape), psf=o.gaussian_psf((5, 5, 5), (1.2, 1.0, 1.0))[0], density=1e-3)
    plan = RichardsonLucyPlan(vshape, psf, device)
    assert plan.path == "separable (long z, 4 launches)" and plan.separable and not plan.fused
    x = plan(_t(y, device), iterations=5, stats=True)
    _close(x.cpu().numpy(), o.richardson_lucy(y, psf, 5), 5e-5, 2e-5)
    want = o.rl_iteration_scalars(y, psf, 5)
    for name in ("flux", "change", "total"):
        np.testing.assert_allclose(getattr(plan.last_stats, name), want[name], rtol=1e-5, err_msg=name)
    assert torch.equal(x, plan(_t(y, device), iterations=5))
    ypad = plan.new_padded_input()
    ypad.view.copy_(_t(y, device))
    assert torch.equal(plan(ypad, iterations=5), x)
    x0 = np.full(vshape, float(y.mean()), np.float32)
    _close(plan(_t(y, device), iterations=2, x0=_t(x0, device)).cpu().numpy(), o.richardson_lucy(y, psf, 2, x0=x0), 5e-5, 2e-5)
    by_factors = RichardsonLucyPlan(vshape, None, device, psf_factors=factors)
    # (the plan factored the PSF itself -- an SVD: the same kernels to the last bit or two)
    _close(by_factors(_t(y, device), iterations=3).cpu().numpy(), plan(_t(y, device), iterations=3).cpu().numpy(), 2e-5, 1e-6)
    for thin in ((3, 9, 40), (1, 5, 7), (pshape[0] - 1, 20, 66)):
        yt = o.bead_scene(thin, seed=3, psf=None, density=5e-3) if thin[0] > 1 else (np.random.default_rng(1).random(thin) * 50 + 1).astype(np.float32)
        got = RichardsonLucyPlan(thin, psf, device)(_t(yt, device), iterations=2)
        _close(got.cpu().numpy(), o.richardson_lucy(yt, psf, 2), 5e-5, 2e-5)
    with pytest.raises(ValueError, match="must be separable"):
        rot = psf.copy()
        rot[0, 0, 0] += 0.01
        RichardsonLucyPlan(vshape, rot / rot.sum(), device)


def _measured_like_psf(shape, seed):
    """A dense, asymmetric, non-separable PSF: a tilted Gaussian with a weak off-axis lobe, normalised to sum 1."""
    rng = np.random.default_rng(seed)
    z, y, x = np.meshgrid(*[np.arange(n) - n // 2 for n in shape], indexing="ij")
    a = math.radians(25.0)
    zr, xr = math.cos(a) * z + math.sin(a) * x, -math.sin(a) * z + math.cos(a) * x
    s = [max(n / 5.0, 0.8) for n in shape]
    w = np.exp(-0.5 * ((zr / s[0]) ** 2 + (y / s[1]) ** 2 + (xr / s[2]) ** 2))
    w += 0.05 * np.exp(-0.5 * (((z - 1) / s[0]) ** 2 + ((y + 2) / s[1]) ** 2 + ((x - 2) / (0.7 * s[2])) ** 2))
    w *= 1.0 + 0.02 * rng.standard_normal(shape)
    w = np.clip(w, 0.0, None)
    return (w / w.sum()).astype(np.float32)


@pytest.mark.parametrize("vshape,pshape", [((20, 40, 52), (13, 15, 17)), ((7, 9, 11), (3, 5, 3)), ((12, 33, 131), (21, 19, 9)),
                                           ((5, 6, 9), (9, 13, 17)), ((30, 17, 64), (1, 1, 31)), ((1, 35, 86), (1, 9, 25)),
                                           ((2, 1, 3), (3, 1, 1))])
def test_rl_in_the_fourier_domain_vs_oracle(device, vshape, pshape, monkeypatch):
    """Dense PSFs beyond the stencil kernels (csrc/rfft_rows.hip, zcorr.hip, shrimpy_amd/deconvolve_fft.py): the
    iteration with both convolutions as products of spectra, against the oracle's DIRECT stencil (the definition) within
    the RL bar and against its use_fft form; reduction scalars to 1e-5; odd widths, volumes thinner than the PSF,
    an explicit x0, in-place output, repeatability."""
    import torch

    from shrimpy_amd.deconvolve import make_plan
    from shrimpy_amd.deconvolve_fft import FftRichardsonLucyPlan, fft_grid

    psf = _measured_like_psf(pshape, sum(pshape))
    y = o.bead_scene(vshape, seed=sum(vshape), psf=o.gaussian_psf((5, 5, 5), (1.2, 1.0, 1.0))[0], density=2e-3)
    plan = make_plan(vshape, psf, device, method="fft")
    assert isinstance(plan, FftRichardsonLucyPlan) and plan.path == "fft" and not plan.padded_input
    assert all(g >= max(n + p // 2, p) for g, n, p in zip(plan.grid, vshape, pshape)) and plan.grid == fft_grid(vshape, pshape)
    yd = _t(y, device)
    x = plan(yd, iterations=5, stats=True)
    chained_stats = plan.last_stats
    ref = o.richardson_lucy(y, psf, 5)
    _close(x.cpu().numpy(), ref, 2e-4, 1e-4)
    _close(x.cpu().numpy(), o.richardson_lucy(y, psf, 5, use_fft=True), 2e-4, 1e-4)
    assert float(x.min()) >= 0 and bool(torch.isfinite(x).all())
    want = o.rl_iteration_scalars(y, psf, 5)
    for name in ("flux", "change", "total"):
        np.testing.assert_allclose(getattr(plan.last_stats, name), want[name], rtol=1e-5, err_msg=name)
    assert torch.equal(x, plan(yd, iterations=5))                      # no atomics on the data path: bit-repeatable
    assert torch.equal(yd, _t(y, device))                              # y is read, never written
    x0 = np.full(vshape, float(y.mean()), np.float32)
    out = torch.empty(vshape, device=device)
    got = plan(yd, iterations=2, x0=_t(x0, device), out=out)
    assert got.data_ptr() == out.data_ptr()
    _close(out.cpu().numpy(), o.richardson_lucy(y, psf, 2, x0=x0), 5e-5, 2e-5)
    one = plan(yd, iterations=1)
    _close(one.cpu().numpy(), o.richardson_lucy(y, psf, 1), 2e-5, 5e-6)
    # the chained form (LSR_FFT_RL_CHAIN=1: inverse x leg -> epilogue -> forward x leg in one kernel, the ratio never in
    # memory) does the arithmetic of the ten-launch form: the same bits, scalars included
    monkeypatch.setenv("LSR_FFT_RL_CHAIN", "1")
    assert torch.equal(plan(yd, iterations=5, stats=True), x)
    np.testing.assert_allclose(plan.last_stats.flux, chained_stats.flux, rtol=1e-12)


def test_rl_method_auto_sends_large_dense_psfs_to_the_fourier_domain(device):
    """make_plan / richardson_lucy(method=): tuned stencils where they take the PSF, the Fourier-domain iteration for
    dense PSFs beyond them (also beyond 15 taps, which the stencil plan refuses), "direct" / "fft" on request; an
    all-zero stack stays zero (/root/reference/shrimpy/tests/test_mantis_integration.py:285-341); tol stops early."""
    import torch

    from shrimpy_amd.deconvolve import RichardsonLucyPlan, make_plan, richardson_lucy

    vshape = (16, 30, 44)
    sep, _ = o.gaussian_psf((9, 7, 7), (2.0, 1.2, 1.2))
    assert make_plan(vshape, sep, device).path == "fused"
    assert make_plan(vshape, o.rotated_psf((9, 7, 7)), device).path.startswith("y-separable")
    mid = _measured_like_psf((13, 13, 13), 1)
    assert RichardsonLucyPlan(vshape, mid, device).path == "generic"
    assert make_plan(vshape, mid, device).path == "fft" and make_plan(vshape, mid, device, method="direct").path == "generic"
    big = _measured_like_psf((17, 19, 19), 2)
    with pytest.raises(ValueError, match="exceeds 15 taps"):
        make_plan(vshape, big, device, method="direct")
    assert make_plan(vshape, big, device).path == "fft"
    assert make_plan(vshape, sep, device, method="fft").path == "fft"
    y = o.bead_scene(vshape, seed=11, psf=sep, density=2e-3)
    yd = _t(y, device)
    # the two routes agree with each other where both exist
    a = richardson_lucy(yd, mid, iterations=4, method="fft")
    b = richardson_lucy(yd, mid, iterations=4, method="direct")
    _close(a.cpu().numpy(), b.cpu().numpy(), 2e-4, 1e-4)
    _close(richardson_lucy(yd, big, iterations=3).cpu().numpy(), o.richardson_lucy(y, big, 3), 2e-4, 1e-4)
    zeros = richardson_lucy(torch.zeros(vshape, device=device), big, iterations=20)
    assert float(zeros.abs().max()) == 0.0
    full, st_full = richardson_lucy(yd, big, iterations=12, return_stats=True)
    x, st = richardson_lucy(yd, big, iterations=12, tol=float(st_full.rel_change[3]) * 1.0001, return_stats=True)
    assert st.stopped_by_tol and 4 <= st.iterations <= 6 and st.iterations < 12
    np.testing.assert_allclose(st.flux, st_full.flux[:st.iterations], rtol=1e-12)


def test_reconstructor_takes_a_measured_psf_through_the_fourier_domain(device, tmp_path):
    """DeconvolveSettings.psf_path with a bead patch of the size scripts/measure_psf.py:187-190 cuts (15 x 18 x 18:
    even extents are padded to odd): the pipeline object plans the Fourier-domain iteration and its output is the
    oracle's chain."""
    from shrimpy_amd.pipeline import VolumeReconstructor
    from shrimpy_amd.settings import DeconvolveSettings, DeskewSettings, ReconstructSettings

    patch = _measured_like_psf((15, 19, 19), 5)[:, :18, :18]
    patch = (patch / patch.sum()).astype(np.float32)
    np.save(tmp_path / "beads.npy", patch)
    raw = o.bead_scene((40, 12, 16), seed=9, psf=None, density=5e-3)
    desk = DeskewSettings(pixel_size_um=0.1133, ls_angle_deg=30.0, scan_step_um=0.15, keep_overhang=False, average_n_slices=3)
    settings = ReconstructSettings(deskew=desk, deconvolution=DeconvolveSettings(psf_path=str(tmp_path / "beads.npy"), iterations=4))
    rec = VolumeReconstructor(raw.shape, settings, device)
    assert rec._plan.path == "fft" and rec._plan.psf.shape == (15, 19, 19)
    got = rec(raw).cpu().numpy()
    ref = o.richardson_lucy(o.deskew(raw, 30.0, 0.755, False, 3), np.pad(patch, ((0, 0), (0, 1), (0, 1))), 4)
    assert got.shape == ref.shape
    _close(got, ref, 2e-4, 1e-4)


def test_correlate_z_entry_against_scipy_on_ragged_and_strided_volumes(device):
    """lsr_correlate_z_f32: every odd tap count up to 31, widths that are not multiples of four, padded rows, many z
    chunks -- against scipy.ndimage.correlate1d(mode="constant")."""
    import ctypes

    import torch

    from scipy import ndimage

    from shrimpy_amd import _lib

    rng = np.random.default_rng(8)
    for pz, shape in ((31, (70, 9, 37)), (17, (5, 3, 6)), (1, (4, 4, 4)), (23, (300, 2, 130)), (29, (33, 17, 1027))):
        w = rng.random(pz).astype(np.float32)
        vol = rng.random(shape).astype(np.float32)
        z, y, x = shape
        pitch = x + 5
        src = torch.zeros((z, y, pitch), device=device)
        src[:, :, :x] = _t(vol, device)
        out = torch.empty(shape, device=device)
        wd = _t(w, device)
        _lib.call("lsr_correlate_z_f32", src.data_ptr(), pitch, y * pitch, None, 0, 0, out.data_ptr(), x, y * x, z, y, x,
                  wd.data_ptr(), pz, _lib.EPI_NONE, ctypes.c_float(0.0), None, None, None, None, _lib.stream_ptr(device))
        np.testing.assert_allclose(out.cpu().numpy(), ndimage.correlate1d(vol, w, axis=0, mode="constant"), rtol=2e-6, atol=1e-6)
    assert _lib.call_value("lsr_correlate_z_max_taps") == 31
    with pytest.raises(_lib.LsrUnsupported):
        _lib.call("lsr_correlate_z_f32", src.data_ptr(), pitch, y * pitch, None, 0, 0, out.data_ptr(), x, y * x, z, y, x,
                  wd.data_ptr(), 33, _lib.EPI_NONE, ctypes.c_float(0.0), None, None, None, None, _lib.stream_ptr(device))


def test_rl_edge_cases(device):
    import torch

    from shrimpy_amd.deconvolve import richardson_lucy

    psf, _ = o.gaussian_psf((5, 5, 5), (1.0, 1.0, 1.0))
    # all-zero (autofocus-failed) volume: stays zero, no NaN
    x = richardson_lucy(torch.zeros((6, 10, 12), device=device), psf, iterations=4)
    assert torch.all(x == 0)
    # volume thinner than the PSF on every axis
    rng = np.random.default_rng(43)
    y = (rng.random((3, 4, 2)) * 50 + 1).astype(np.float32)
    _close(richardson_lucy(_t(y, device), psf, iterations=3).cpu().numpy(),
           o.richardson_lucy(y, psf, 3), 5e-5, 2e-5)
    _close(richardson_lucy(_t(y, device), psf, iterations=3, separable="never").cpu().numpy(),
           o.richardson_lucy(y, psf, 3), 5e-5, 2e-5)
    # delta PSF is the identity; zero iterations returns x0 = y
    delta = np.zeros((3, 3, 3), np.float32)
    delta[1, 1, 1] = 1
    yy = _t((rng.random((5, 9, 70)) * 10 + 1).astype(np.float32), device)
    torch.testing.assert_close(richardson_lucy(yy, delta, iterations=3), yy, rtol=1e-5, atol=0)
    assert torch.equal(richardson_lucy(yy, psf, iterations=0), yy)
    # even-sized PSF axes are zero-padded, same answer as the oracle
    w = (rng.random((2, 4, 3)) + 0.1).astype(np.float32)
    w /= w.sum()
    y2 = (rng.random((6, 12, 14)) * 50 + 1).astype(np.float32)
    _close(richardson_lucy(_t(y2, device), w, iterations=3).cpu().numpy(), o.richardson_lucy(y2, w, 3), 5e-5, 2e-5)


# ================================================================ size-independent properties at scale


def test_deskew_properties_at_larger_size(device):
    """(i) linearity; (ii) exact reproduction of a field linear in z (order-1 is exact on it)."""
    import torch

    from shrimpy_amd.deskew import fast_deskew_zyx
    from shrimpy_amd.geometry import deskew_geometry

    Z, Y, X = 640, 96, 320
    kw = dict(ls_angle_deg=30.0, px_to_scan_ratio=0.755, keep_overhang=False, average_n_slices=1)
    g = torch.Generator(device=device).manual_seed(5)
    a = torch.rand((Z, Y, X), device=device, generator=g)
    b = torch.rand((Z, Y, X), device=device, generator=g)
    da, db = fast_deskew_zyx(raw_data=a, **kw), fast_deskew_zyx(raw_data=b, **kw)
    dab = fast_deskew_zyx(raw_data=a + 2 * b, **kw)
    torch.testing.assert_close(dab, da + 2 * db, rtol=1e-5, atol=1e-5)

    ramp = torch.arange(Z, device=device, dtype=torch.float32)[:, None, None].expand(Z, Y, X).contiguous()
    out = fast_deskew_zyx(raw_data=ramp, **kw)
    geo = deskew_geometry((Z, Y, X), 30.0, 0.755, False)
    m = geo.matrix_3x4
    zp = torch.arange(out.shape[0], device=device, dtype=torch.float64)[:, None, None]
    xp = torch.arange(out.shape[2], device=device, dtype=torch.float64)[None, None, :]
    z_in = (zp * m[0, 0] + xp * m[0, 2] + m[0, 3]).expand(out.shape)
    inside = (z_in >= 0) & (z_in <= Z - 1)
    assert float((out.double() - z_in)[inside].abs().max()) < 1e-3
    assert float(out[~inside].abs().max()) == 0


def test_rl_properties_at_larger_size(device):
    """Non-negativity, finiteness, fixed point on a constant interior, sep == dense."""
    import torch

    from shrimpy_amd.deconvolve import RichardsonLucyPlan

    psf, _ = o.gaussian_psf((9, 7, 7), (2.0, 1.2, 1.2))
    shape = (60, 200, 300)
    g = torch.Generator(device=device).manual_seed(7)
    y = torch.poisson(100 + 3000 * (torch.rand(shape, device=device, generator=g) > 0.9995).float())
    sep = RichardsonLucyPlan(shape, psf, device)(y, iterations=10)
    dense = RichardsonLucyPlan(shape, psf, device, separable="never")(y, iterations=10)
    assert torch.isfinite(sep).all() and float(sep.min()) >= 0
    torch.testing.assert_close(sep, dense, rtol=2e-4, atol=1e-4 * float(dense.max()))
    # a constant volume is a fixed point wherever two PSF radii separate the voxel from a border
    c = torch.full(shape, 37.0, device=device)
    out = RichardsonLucyPlan(shape, psf, device)(c, iterations=3)
    inner = (slice(8 * 3, -8 * 3), slice(6 * 3, -6 * 3), slice(6 * 3, -6 * 3))
    torch.testing.assert_close(out[inner], c[inner], rtol=2e-5, atol=0)


def test_deskew_into_padded_volume_and_rl_without_copies(device):
    """The pipeline's fast route: the deskew kernel writes its output into the RL plan's padded,
    line-aligned volume (same bits as the dense output, halo untouched = zero), and RL starts from
    x0 = y there without the pad / init copies.  Same answer as the dense route."""
    import torch

    from shrimpy_amd.deconvolve import RichardsonLucyPlan
    from shrimpy_amd.deskew import deskew_with_matrix
    from shrimpy_amd.geometry import deskew_geometry

    psf, factors = o.gaussian_psf((9, 7, 7), (2.0, 1.2, 1.2))
    raw = o.bead_scene((140, 30, 100), seed=91, psf=psf, density=1e-3)
    geo = deskew_geometry(raw.shape, 30.0, 0.755, False, 3)
    raw_t = _t(raw, device)
    dense = deskew_with_matrix(raw_t, geo.matrix_3x4, geo.pre_average_shape, 3)
    for kw in (dict(psf_factors=factors), dict(separable="never")):
        plan = RichardsonLucyPlan(geo.output_shape, None if "psf_factors" in kw else o.rotated_psf(), device, **kw)
        ypad = plan.new_padded_input()
        got = deskew_with_matrix(raw_t, geo.matrix_3x4, geo.pre_average_shape, 3, out=ypad)
        assert got is ypad and torch.equal(ypad.view, dense)
        mask = torch.ones_like(ypad.full, dtype=torch.bool)
        _, _, rows, oy, ox = plan.padded_geometry()
        mask[:, oy:oy + dense.shape[1], ox:ox + dense.shape[2]] = False
        assert float(ypad.full[mask].abs().max()) == 0.0          # the kernel never wrote the halo
        a = plan(ypad, iterations=6)
        b = plan(dense, iterations=6)
        torch.testing.assert_close(a, b, rtol=1e-6, atol=0)
    _close(a.cpu().numpy(), o.richardson_lucy(dense.cpu().numpy(), o.rotated_psf(), 6), 5e-5, 2e-5)


# ================================================================ flat-field


def _median_torch_semantics(vol):
    """Exact middle elements; torch.quantile's lerp between the two of an even count (f32)."""
    s = np.sort(vol, axis=0)
    z = vol.shape[0]
    a, b = s[(z - 1) // 2], s[z // 2]
    return a if z % 2 else (b - (b - a) * np.float32(0.5)).astype(np.float32)


@pytest.mark.parametrize("shape", [(8, 6, 10), (7, 5, 9), (300, 3, 130), (64, 17, 200), (1, 4, 5), (2, 3, 129),
                                   (1201, 2, 70)])
@pytest.mark.parametrize("kind", ["counts", "floats", "ties"])
def test_flatfield_pattern_is_the_exact_median(device, shape, kind):
    """Radix select returns the exact order statistics: the pattern equals sort-based medians bit
    for bit (integer camera counts with many ties, signed floats, near-constant columns)."""
    from shrimpy_amd.flatfield import flat_field_pattern

    rng = np.random.default_rng(hash((shape, kind)) % 2**32)
    if kind == "counts":
        vol = rng.integers(80, 600, shape).astype(np.float32)
    elif kind == "floats":
        vol = (rng.standard_normal(shape) * 100).astype(np.float32)
        vol[0, 0, 0] = -0.0
    else:
        vol = np.full(shape, 37.5, np.float32)
        vol[rng.random(shape) < 0.3] = 37.500004
        vol[rng.random(shape) < 0.1] = -1e30
    ff = flat_field_pattern(_t(vol, device))
    pattern = ff.pattern.cpu().numpy()
    expect = _median_torch_semantics(vol)
    np.testing.assert_array_equal(pattern, expect)
    assert float(ff.mean.cpu()[0]) == pytest.approx(float(expect.astype(np.float64).mean()), rel=1e-6)
    if kind == "counts":
        out = ff.apply(_t(vol, device)).cpu().numpy()
        mean = ff.mean.cpu().numpy()[0]
        np.testing.assert_array_equal(out, vol / pattern * mean)     # same operations, same order
        np.testing.assert_allclose(out, o.flat_field_bf(vol), rtol=2e-6)


def test_flatfield_count_fast_path_falls_back_when_a_late_sample_is_not_a_count(device):
    """Stacks of integer camera counts take the two-pass 16-bit select; a workgroup that meets a
    non-count later in z (fraction, negative, >= 65536, NaN) restarts with the float keys."""
    from shrimpy_amd.flatfield import flat_field_pattern

    rng = np.random.default_rng(21)
    base = rng.integers(0, 65536, (33, 3, 300)).astype(np.float32)
    np.testing.assert_array_equal(flat_field_pattern(_t(base, device)).pattern.cpu().numpy(),
                                  _median_torch_semantics(base))
    for bad in (0.5, -3.0, 65536.0, 1e9):
        vol = base.copy()
        vol[31, 1, 7] = bad            # first workgroup (pixels 0..127 of row 0 ... ) restarts
        vol[17, 2, 299] = bad          # and the last one
        np.testing.assert_array_equal(flat_field_pattern(_t(vol, device)).pattern.cpu().numpy(),
                                      _median_torch_semantics(vol))


def test_flatfield_nan_propagates_per_pixel(device):
    from shrimpy_amd.flatfield import flat_field_pattern

    rng = np.random.default_rng(12)
    vol = rng.random((9, 4, 70)).astype(np.float32) + 1
    vol[3, 2, 5] = np.nan
    pattern = flat_field_pattern(_t(vol, device)).pattern.cpu().numpy()
    assert np.isnan(pattern[2, 5]) and np.isnan(pattern).sum() == 1
    clean = np.delete(pattern.ravel(), 2 * 70 + 5)
    np.testing.assert_array_equal(clean, np.delete(_median_torch_semantics(vol).ravel(), 2 * 70 + 5))


def test_flatfield_fused_into_deskew_is_bit_identical(device):
    """in / pattern * mean applied inside the deskew kernel == apply, then deskew."""
    import torch

    from shrimpy_amd.deskew import deskew_with_matrix
    from shrimpy_amd.flatfield import flat_field_pattern
    from shrimpy_amd.geometry import deskew_geometry

    rng = np.random.default_rng(77)
    raw = _t(rng.integers(80, 600, (150, 20, 90)).astype(np.float32), device)
    geo = deskew_geometry(tuple(raw.shape), 30.0, 0.755, False, 3)
    ff = flat_field_pattern(raw)
    two_step = deskew_with_matrix(ff.apply(raw), geo.matrix_3x4, geo.pre_average_shape, 3)
    fused = deskew_with_matrix(raw, geo.matrix_3x4, geo.pre_average_shape, 3, flat_field=ff)
    assert torch.equal(fused, two_step)
    ref = o.deskew(o.flat_field_bf(raw.cpu().numpy()), 30.0, 0.755, False, 3)
    np.testing.assert_allclose(fused.cpu().numpy(), ref, rtol=3e-6, atol=1e-4)


def test_preprocessor_flatfield_then_deskew_on_the_gpu(device):
    from shrimpy_amd.preprocessing import build_preprocessor

    rng = np.random.default_rng(5)
    raw = rng.integers(80, 600, (120, 12, 64)).astype(np.float32)
    settings = dict(pixel_size_um=0.1133, ls_angle_deg=30.0, px_to_scan_ratio=0.755, keep_overhang=False,
                    average_n_slices=3)
    pre = build_preprocessor(raw.shape, ["flatfield", "deskew"], deskew=settings)
    out = pre(raw, label="A/1/0", return_intermediates=True)
    ref = o.deskew(o.flat_field_bf(raw), 30.0, 0.755, False, 3)
    np.testing.assert_allclose(out["deskew"].cpu().numpy(), ref, rtol=3e-6, atol=1e-4)
    only = build_preprocessor(raw.shape, ["flatfield"])(raw)
    np.testing.assert_allclose(next(iter(only.values())).cpu().numpy(), o.flat_field_bf(raw), rtol=2e-6)


def test_deskew_uint16_stack_equals_the_float_path(device):
    """Camera counts as uint16 in, float32 out: the same bits as deskewing the float32 copy, into
    a dense tensor and into a padded RL input; the preprocessor uploads uint16 stacks unconverted."""
    import torch

    from shrimpy_amd.deconvolve import RichardsonLucyPlan
    from shrimpy_amd.deskew import deskew_with_matrix, fast_deskew_zyx
    from shrimpy_amd.geometry import deskew_geometry
    from shrimpy_amd.preprocessing import build_preprocessor

    rng = np.random.default_rng(8)
    raw = rng.integers(0, 65536, (150, 20, 90)).astype(np.uint16)
    kw = dict(ls_angle_deg=30.0, px_to_scan_ratio=0.755, keep_overhang=False, average_n_slices=3)
    as_f32 = fast_deskew_zyx(raw_data=_t(raw.astype(np.float32), device), **kw)
    as_u16 = fast_deskew_zyx(raw_data=torch.as_tensor(raw, device=device), **kw)
    assert as_u16.dtype == torch.float32 and torch.equal(as_u16, as_f32)
    geo = deskew_geometry(raw.shape, 30.0, 0.755, False, 3)
    _, factors = o.gaussian_psf((9, 7, 7), (2.0, 1.2, 1.2))
    ypad = RichardsonLucyPlan(geo.output_shape, None, device, psf_factors=factors).new_padded_input()
    deskew_with_matrix(torch.as_tensor(raw, device=device), geo.matrix_3x4, geo.pre_average_shape, 3, out=ypad)
    assert torch.equal(ypad.view, as_f32)
    pre = build_preprocessor(raw.shape, ["deskew"], deskew=dict(pixel_size_um=0.1133, **kw))
    out = pre(raw)
    assert torch.equal(next(iter(out.values())), as_f32)


def test_flatfield_and_fused_deskew_on_uint16_stacks(device):
    """The uint16 variants give the bits of the float32 path on the converted stack."""
    import torch

    from shrimpy_amd.deskew import deskew_with_matrix
    from shrimpy_amd.flatfield import flat_field_bf, flat_field_pattern
    from shrimpy_amd.geometry import deskew_geometry
    from shrimpy_amd.preprocessing import build_preprocessor

    rng = np.random.default_rng(14)
    raw = rng.integers(80, 600, (150, 20, 90)).astype(np.uint16)
    r16, r32 = torch.as_tensor(raw, device=device), _t(raw.astype(np.float32), device)
    f16, f32 = flat_field_pattern(r16), flat_field_pattern(r32)
    assert torch.equal(f16.pattern, f32.pattern) and torch.equal(f16.mean, f32.mean)
    np.testing.assert_array_equal(f16.pattern.cpu().numpy(), _median_torch_semantics(raw.astype(np.float32)))
    assert torch.equal(flat_field_bf(r16), flat_field_bf(r32))
    geo = deskew_geometry(raw.shape, 30.0, 0.755, False, 3)
    a = deskew_with_matrix(r16, geo.matrix_3x4, geo.pre_average_shape, 3, flat_field=f16)
    b = deskew_with_matrix(r32, geo.matrix_3x4, geo.pre_average_shape, 3, flat_field=f32)
    assert torch.equal(a, b)
    settings = dict(pixel_size_um=0.1133, ls_angle_deg=30.0, px_to_scan_ratio=0.755, keep_overhang=False,
                    average_n_slices=3)
    pre = build_preprocessor(raw.shape, ["flatfield", "deskew"], deskew=settings)
    assert torch.equal(next(iter(pre(raw).values())), b)


def test_rl_y_separable_psf_takes_the_stencil_plus_y_pass_and_matches_dense(device):
    """A PSF tilted in (z, x) and Gaussian along y (the oblique light sheet's, and the rotated PSF of
    the secondary benchmark) factors as ky (x) kzx: the plan then runs a (z, x) stencil and a y pass
    per correlation.  Same result as the 441-tap dense kernel and the oracle, within the RL bar."""
    import torch

    from shrimpy_amd.deconvolve import RichardsonLucyPlan, factor_psf_y

    psf = o.rotated_psf((9, 7, 7), (2.0, 1.2, 1.2), 30.0)
    ky, kzx = factor_psf_y(psf)
    assert ky.shape == (7,) and kzx.shape == (9, 7) and abs(float(ky.sum()) - 1.0) < 1e-6
    shape = (21, 70, 150)
    rng = np.random.default_rng(44)
    y = (rng.poisson(100 + 3000 * (rng.random(shape) > 0.999)).astype(np.float32))
    auto = RichardsonLucyPlan(shape, psf, device)                 # the default: one launch per iteration (round 4)
    two = RichardsonLucyPlan(shape, psf, device, fused="never")
    dense = RichardsonLucyPlan(shape, psf, device, separable="never")
    assert (auto.path, two.path, dense.path) == ("y-separable (fused)", "y-separable", "dense")
    a, b = auto(_t(y, device), iterations=8), dense(_t(y, device), iterations=8)
    ref = o.richardson_lucy(y, psf, 8)
    for got in (a, b):
        _close(got.cpu().numpy(), ref, 2e-4, 1e-4)
    # one launch per iteration == one launch per correlation, bit for bit (odd and even iteration counts:
    # the result sits in either working volume)
    for n in (1, 2, 8):
        assert torch.equal(auto(_t(y, device), iterations=n), two(_t(y, device), iterations=n)), n
    # x0, a padded y written by a producer, zero iterations, one-plane and thin volumes
    x0 = _t(np.full(shape, 50.0, np.float32), device)
    _close(auto(_t(y, device), iterations=3, x0=x0).cpu().numpy(), o.richardson_lucy(y, psf, 3, x0=x0.cpu().numpy()),
           2e-4, 1e-4)
    ypad = auto.new_padded_input()
    ypad.view.copy_(_t(y, device))
    assert torch.equal(auto(ypad, iterations=8), a)
    for thin in ((1, 9, 40), (3, 2, 5), (12, 33, 65)):
        yt = (rng.random(thin) * 50 + 1).astype(np.float32)
        p = RichardsonLucyPlan(thin, psf, device, fused="always")
        assert p.path == "y-separable (fused)"
        got = p(_t(yt, device), iterations=2)
        _close(got.cpu().numpy(), o.richardson_lucy(yt, psf, 2), 2e-4, 1e-4)
        assert torch.equal(got, RichardsonLucyPlan(thin, psf, device, fused="never")(_t(yt, device), iterations=2))


@pytest.mark.parametrize("seed", [45, 46])
def test_rl_y_separable_random_psf_shapes(device, seed):
    """The fused ky (x) kzx kernel on random PSF extents and awkward volumes against the oracle and, bit for bit, against
    the two-launch form.  (Rounds 3-4 also had a 256-thread shape on 32 x 64 tiles, selected by ``LSR_YSEP_SHAPE``: the
    slower one once the border normalisation was fixed, and one of its instances faulted with the scalar-load taps --
    removed; the variable is ignored.)"""
    import torch

    from shrimpy_amd.deconvolve import RichardsonLucyPlan

    rng = np.random.default_rng(seed)
    for case in range(10):
        pz, py, px = int(rng.choice([3, 5, 7, 9, 11])), int(rng.choice([3, 5, 9, 13, 15])), int(rng.choice([3, 5, 7, 9]))
        kzx = np.abs(rng.normal(1.0, 0.5, (pz, px))) + 0.05
        ky = np.abs(rng.normal(1.0, 0.4, py)) + 0.05
        psf = (ky[None, :, None] * kzx[:, None, :]).astype(np.float32)
        psf /= psf.sum()
        shape = (int(rng.integers(1, 30)), int(rng.integers(1, 80)), int(rng.integers(1, 200)))
        y = (rng.random(shape) * 80 + 1).astype(np.float32)
        plan = RichardsonLucyPlan(shape, psf, device, fused="always")
        assert plan.path == ("y-separable (fused)" if py <= 9 else "y-separable (4 launches)"), (case, psf.shape)
        iters = int(rng.integers(1, 4))
        got = plan(_t(y, device), iterations=iters)
        _close(got.cpu().numpy(), o.richardson_lucy(y, psf, iters), 2e-4, 1e-4)
        if py <= 9:   # every compiled extent: the fused iteration is the two-launch form, bit for bit
            two = RichardsonLucyPlan(shape, psf, device, fused="never")
            assert two.path == "y-separable"
            assert torch.equal(got, two(_t(y, device), iterations=iters)), (case, psf.shape, shape)


def test_zxy_entry_rejects_what_it_does_not_cover(device):
    """The one-launch ky (x) kzx entry carries the RL epilogues only, needs its y taps, and shares the
    dense kernel's tap range; errors come back as status codes with a message, nothing is launched."""
    import ctypes

    import torch

    from shrimpy_amd import _lib
    from shrimpy_amd.deconvolve import PaddedVolume

    shape, pshape = (6, 20, 40), (5, 5, 5)
    a, b = PaddedVolume(shape, pshape, device), PaddedVolume(shape, pshape, device)
    taps = torch.zeros(_lib.call_value("lsr_dense_taps_count", *pshape), device=device)
    ky = torch.ones(5, device=device) / 5
    table = torch.zeros(6 * 6 * 6, dtype=torch.float64, device=device)
    z, y, x = shape

    def call(epi, ky_ptr=ky.data_ptr(), p=pshape, table_ptr=table.data_ptr()):
        _lib.call("lsr_correlate_zxy_padded_f32", a.logical_ptr(), a.pitch, a.plane, a.logical_ptr(), a.pitch, a.plane,
                  b.logical_ptr(), b.pitch, b.plane, z, y, x, taps.data_ptr(), ky_ptr, *p, epi, ctypes.c_float(1e-6),
                  table_ptr, ctypes.c_float(1.0), _lib.stream_ptr(device))

    with pytest.raises(_lib.LsrError, match="RL epilogues only"):
        call(_lib.EPI_NONE)
    with pytest.raises(_lib.LsrError):
        call(_lib.EPI_RATIO, ky_ptr=None)
    with pytest.raises(_lib.LsrError):
        call(_lib.EPI_UPDATE, table_ptr=None)
    with pytest.raises(_lib.LsrUnsupported):
        call(_lib.EPI_RATIO, p=(5, 11, 5))      # y taps beyond the dense kernel's nine
    call(_lib.EPI_RATIO)                         # and the valid call goes through
    torch.cuda.synchronize()


def test_registration_writes_the_rl_input_in_place(device):
    """deskew -> register -> RL through ``VolumeReconstructor``: the registration's result lands in the RL
    plan's padded input (``apply_affine_transform_zyx(out=PaddedVolume)``, strided stores of all three
    affine kernels) -- same bits as the dense call, and the chain follows the oracle."""
    import torch

    from shrimpy_amd.deconvolve import RichardsonLucyPlan
    from shrimpy_amd.pipeline import VolumeReconstructor
    from shrimpy_amd.register import apply_affine_transform_zyx
    from shrimpy_amd.settings import DeconvolveSettings, DeskewSettings, ReconstructSettings, RegisterSettings

    rng = np.random.default_rng(37)
    psf, _ = o.gaussian_psf((9, 7, 7), (2.0, 1.2, 1.2))
    for shape, m, mode in [((20, 96, 132), _config3_matrix(), "constant"),                                   # planar
                           ((24, 96, 132), _tilted_matrix([(1, 3.0), (0, 2.0)], (1.0, 0.98, 1.02), (1.5, -4.25, 6.75)), "constant"),
                           ((20, 50, 70), _tilted_matrix([(1, 3.0)]), "grid-constant")]:                    # gather (Xi % 4)
        vol = (rng.random(shape) * 1000).astype(np.float32)
        dense = apply_affine_transform_zyx(_t(vol, device), m, shape, mode=mode)
        plan = RichardsonLucyPlan(shape, psf, device)
        pad = plan.new_padded_input()
        got = apply_affine_transform_zyx(_t(vol, device), m, shape, mode=mode, out=pad)
        assert got is pad and torch.equal(pad.view, dense)
        halo = pad.full.clone()
        halo[:, pad.view.storage_offset() % pad.plane // pad.pitch:, :][:, :shape[1], pad.view.storage_offset() % pad.pitch:][:, :, :shape[2]] = 0
        assert not halo.any()                                  # nothing written outside the window
    raw = rng.integers(80, 600, (98, 24, 40)).astype(np.uint16)
    d = DeskewSettings(pixel_size_um=0.1133, scan_step_um=0.15, ls_angle_deg=30.0, keep_overhang=True, average_n_slices=3)
    m = _tilted_matrix([(1, 2.0), (0, 1.0)], (1.0, 0.99, 1.01), (0.5, -1.25, 2.75))
    rec = VolumeReconstructor(raw.shape, ReconstructSettings(
        deskew=d, registration=RegisterSettings(affine_transform_zyx=m.tolist()),
        deconvolution=DeconvolveSettings(iterations=5)), device)
    desk = o.deskew(raw.astype(np.float32), 30.0, 0.755, True, 3)
    reg = o.affine_apply_4x4(desk, m, desk.shape, cval=0.0, mode="constant")
    _close(rec(raw).cpu().numpy(), o.richardson_lucy(reg, psf, iterations=5), 5e-5, 2e-5)
