"""The native host twins (``csrc/host_twins.hip`` behind ``shrimpy_amd.host``): the product's path for CPU
tensors, i.e. for boxes where the reference itself resolves to ``cpu`` (``shrimpy/preprocessing.py:78-82``)
and for BASELINE config 1.  Checked against the oracle and the golden vectors here (no GPU needed), and
against the device kernels bit for bit in the ``gpu`` tests at the bottom.
"""

import numpy as np
import pytest

from oracle import cpu_ref as o
from shrimpy_amd import _lib

RL_TOL = dict(rel=2e-4, floor=1e-4)      # the RL bar of tests/test_gpu_parity.py


def _t(a):
    import torch

    return torch.as_tensor(np.ascontiguousarray(a))


def _rl_close(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    tol = RL_TOL["rel"] * np.abs(want) + RL_TOL["floor"] * np.abs(want).max()
    assert np.all(np.abs(got - want) <= tol), float(np.max(np.abs(got - want) / tol))


@pytest.mark.parametrize("name", ["deskew_nooverhang_avg3", "deskew_overhang_avg1", "deskew_nooverhang_avg5_r0p4"])
def test_deskew_on_a_cpu_tensor_equals_the_golden_vectors(golden_dir, name):
    from shrimpy_amd.deskew import fast_deskew_zyx

    g = np.load(golden_dir / f"{name}.npz")
    kw = dict(ls_angle_deg=float(g["ls_angle_deg"]), px_to_scan_ratio=float(g["px_to_scan_ratio"]),
              keep_overhang=bool(g["keep_overhang"]), average_n_slices=int(g["average_n_slices"]))
    out = fast_deskew_zyx(raw_data=_t(g["raw"]), **kw)
    assert out.device.type == "cpu"
    np.testing.assert_array_equal(out.numpy(), g["out"])


def test_config1_deskew_through_the_product_path_equals_the_oracle():
    """BASELINE configs[0]: a single 256x256x64 oblique stack (raw (256, 64, 256)), deskew only, no GPU."""
    from shrimpy_amd.deskew import deskew_data, fast_deskew_zyx, get_deskewed_data_shape

    psf_factors = o.gaussian_psf((9, 7, 7), (2.0, 1.2, 1.2))[1]
    raw = o.bead_scene((256, 64, 256), seed=1000, psf_factors=psf_factors)
    kw = dict(ls_angle_deg=30.0, px_to_scan_ratio=0.755, keep_overhang=False, average_n_slices=3)
    want = o.deskew(raw, 30.0, 0.755, False, 3)
    out = fast_deskew_zyx(raw_data=_t(raw), **kw)
    assert tuple(out.shape) == get_deskewed_data_shape(raw.shape, **kw)[0]
    np.testing.assert_array_equal(out.numpy(), want)
    # camera counts go in unconverted, the older numpy entry point and a thread pool give the same bits
    np.testing.assert_array_equal(fast_deskew_zyx(raw_data=_t(raw.astype(np.uint16)), **kw).numpy(),
                                  o.deskew(raw.astype(np.uint16).astype(np.float32), 30.0, 0.755, False, 3))
    import torch

    before = torch.get_num_threads()
    try:
        torch.set_num_threads(4)
        np.testing.assert_array_equal(deskew_data(raw, device="cpu", **kw), want)
        assert _lib.call_value("lsr_get_host_threads") == 4
    finally:
        torch.set_num_threads(before)


@pytest.mark.parametrize("border", ["constant", "grid-constant"])
@pytest.mark.parametrize("keep_overhang,avg", [(True, 1), (False, 3), (True, 4)])
def test_deskew_borders_and_the_general_matrix_route(border, keep_overhang, avg):
    from shrimpy_amd.deskew import deskew_with_matrix, fast_deskew_zyx
    from shrimpy_amd.geometry import deskew_geometry

    raw = np.random.default_rng(3).poisson(300, (70, 12, 18)).astype(np.float32)
    out = fast_deskew_zyx(raw_data=_t(raw), ls_angle_deg=30.0, px_to_scan_ratio=0.755, keep_overhang=keep_overhang,
                          average_n_slices=avg, border=border)
    np.testing.assert_array_equal(out.numpy(), o.deskew(raw, 30.0, 0.755, keep_overhang, avg, border=border))
    # a matrix that is not a deskew shear: trilinear resample, then the average
    geo = deskew_geometry(raw.shape, 30.0, 0.755, keep_overhang, avg)
    m = np.array(geo.matrix_3x4, dtype=np.float64)
    m[1, 2], m[2, 0] = 0.03, -0.02
    got = deskew_with_matrix(_t(raw), m, geo.pre_average_shape, avg, border=border)
    pre = o.affine_apply(raw, m[:, :3], m[:, 3], geo.pre_average_shape, mode=border)
    np.testing.assert_array_equal(got.numpy(), o.average_slices(pre, avg))


@pytest.mark.parametrize("avg", [1, 3])
@pytest.mark.parametrize("cval", [7.25, -3.0, "min", None])
def test_deskew_fill_value_on_a_cpu_tensor_is_scipys_cval(cval, avg):
    """``cval`` (a number, or "min" / None = the stack's minimum: the third [RECALLED] biahub convention, round-4 verdict
    item 6) on the host twin: bit for bit ``scipy.ndimage.affine_transform(cval=...)`` + the slice average, float32 and
    uint16 stacks, the shear kernel and the general-matrix route."""
    import torch

    from shrimpy_amd.deskew import deskew_with_matrix, fast_deskew_zyx
    from shrimpy_amd.geometry import deskew_geometry

    rng = np.random.default_rng(17)
    raw = rng.integers(90, 600, (70, 20, 33)).astype(np.float32)
    for keep in (False, True):
        want = o.deskew(raw, 30.0, 0.755, keep, avg, cval=cval)
        got = fast_deskew_zyx(torch.as_tensor(raw), 30.0, 0.755, keep, avg, cval=cval).numpy()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        got16 = fast_deskew_zyx(torch.as_tensor(raw.astype(np.uint16)), 30.0, 0.755, keep, avg, cval=cval).numpy()
        assert np.array_equal(got16.view(np.uint32), want.view(np.uint32))
    # the grid-constant border has no shear twin on the host: the generic resampler takes the same fill value
    want = o.deskew(raw, 30.0, 0.755, True, avg, border="grid-constant", cval=cval)
    got = fast_deskew_zyx(torch.as_tensor(raw), 30.0, 0.755, True, avg, border="grid-constant", cval=cval).numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    with pytest.raises(ValueError):
        fast_deskew_zyx(torch.as_tensor(raw), 30.0, 0.755, True, avg, cval="max")
    with pytest.raises(ValueError):
        fast_deskew_zyx(torch.as_tensor(raw), 30.0, 0.755, True, avg, cval=float("nan"))


@pytest.mark.parametrize("mode", ["constant", "grid-constant"])
def test_affine_apply_on_a_cpu_tensor_is_scipy_bit_for_bit(golden_dir, mode):
    from shrimpy_amd.register import affine_transform, apply_affine_transform_zyx

    g = np.load(golden_dir / "affine_rot2deg.npz")
    if mode == "constant":
        got = apply_affine_transform_zyx(_t(g["vol"]), g["matrix"])
        np.testing.assert_array_equal(got.numpy(), g["out_constant"])
    else:
        got = apply_affine_transform_zyx(_t(g["vol"]), g["matrix"], tuple(int(v) for v in g["grid_shape"]), mode=mode,
                                         cval=float(g["grid_cval"]))
        np.testing.assert_array_equal(got.numpy(), g["out_grid"])
    rng = np.random.default_rng(11)
    vol = rng.random((9, 21, 30)).astype(np.float32)
    m = np.eye(4)
    m[:3, :3] += rng.normal(0, 0.05, (3, 3))
    m[:3, 3] = [0.7, -2.2, 3.1]
    got = apply_affine_transform_zyx(_t(vol), m, (11, 19, 33), mode=mode, cval=-2.5, exact=False)   # host: always exact
    np.testing.assert_array_equal(got.numpy(), o.affine_apply_4x4(vol, m, (11, 19, 33), cval=-2.5, mode=mode))
    got = affine_transform(_t(vol), m[:3, :3], offset=m[:3, 3], mode=mode)
    np.testing.assert_array_equal(got.numpy(), o.affine_apply_4x4(vol, m, vol.shape, mode=mode))


def test_richardson_lucy_on_a_cpu_tensor_meets_the_rl_bar(golden_dir):
    from shrimpy_amd.deconvolve import correlate3d, richardson_lucy

    g = np.load(golden_dir / "rl_5iter.npz")
    got = richardson_lucy(_t(g["y"]), g["psf_sep"], iterations=5)
    assert got.device.type == "cpu"
    _rl_close(got.numpy(), g["x_sep_5"])
    _rl_close(richardson_lucy(_t(g["y"]), psf_factors=(g["kz"], g["ky"], g["kx"]), iterations=5).numpy(), g["x_sep_5"])
    _rl_close(richardson_lucy(_t(g["y"]), g["psf_rot"], iterations=5).numpy(), g["x_rot_5"])
    _rl_close(richardson_lucy(_t(g["y"]), g["psf_rot"], iterations=1).numpy(), g["x_rot_1"])
    # a rotated (non-separable) PSF takes the dense twin; zero input stays zero (autofocus-failed stacks)
    psf, factors = o.gaussian_psf((5, 5, 7), (1.1, 0.9, 1.4))
    rot = psf.copy()
    rot[0, 0, 0] += 0.01
    rot /= rot.sum()
    y = o.bead_scene((12, 20, 26), seed=2, psf=psf, density=2e-3)
    _rl_close(richardson_lucy(_t(y), rot, iterations=4).numpy(), o.richardson_lucy(y, rot, iterations=4))
    _rl_close(richardson_lucy(_t(y), psf_factors=factors, iterations=4).numpy(),
              o.richardson_lucy_separable(y, factors, iterations=4))
    assert not richardson_lucy(_t(np.zeros((6, 8, 8), np.float32)), psf, iterations=3).numpy().any()
    # a dense PSF beyond the stencil kernels' 15 taps (on a device: the Fourier-domain iteration; here the twins' dense loop)
    rng = np.random.default_rng(5)
    wide = (np.abs(rng.normal(1.0, 0.4, (17, 5, 19))) + 0.05).astype(np.float32)
    wide /= wide.sum()
    ys = o.bead_scene((6, 9, 12), seed=4, psf=None, density=2e-2)
    _rl_close(richardson_lucy(_t(ys), wide, iterations=3).numpy(), o.richardson_lucy(ys, wide, iterations=3))
    # the building block is scipy's correlate(mode="constant")
    from scipy import ndimage

    np.testing.assert_allclose(correlate3d(_t(y), rot).numpy(), ndimage.correlate(y, rot, mode="constant"), rtol=2e-5, atol=1e-3)
    np.testing.assert_allclose(correlate3d(_t(y), weight_factors=factors).numpy(), ndimage.correlate(y, psf, mode="constant"),
                               rtol=2e-5, atol=1e-3)


def test_rl_scalars_on_a_cpu_tensor_match_the_oracle_and_tol_stops_the_loop():
    """Row g of the round-3 verdict on the host twins: flux / change / total per iteration against the oracle's float64
    sums, flux == sum y, early stop on the relative change, zeros for the all-zero stack."""
    import torch

    from shrimpy_amd.deconvolve import richardson_lucy

    psf, factors = o.gaussian_psf((5, 5, 7), (1.1, 0.9, 1.4))
    rot = o.rotated_psf((5, 5, 7), (1.1, 0.9, 1.4), 30.0)
    y = o.bead_scene((12, 20, 26), seed=2, psf=psf, density=2e-3)
    for kernel in (psf, rot):
        want = o.rl_iteration_scalars(y, kernel, 6)
        plain = richardson_lucy(_t(y), kernel, iterations=6)
        x, s = richardson_lucy(_t(y), kernel, iterations=6, return_stats=True)
        assert torch.equal(x, plain) and s.iterations == 6 and not s.stopped_by_tol
        for name in ("flux", "change", "total"):
            np.testing.assert_allclose(getattr(s, name), want[name], rtol=1e-6, err_msg=name)
        np.testing.assert_allclose(s.flux, float(y.astype(np.float64).sum()), rtol=1e-7)
        rel = want["change"] / want["total"]
        tol = float(0.5 * (rel[2] + rel[3]))
        # iteration 4 (index 3) is the first below tol; like the device plans, whose host reads an iteration's scalars while
        # the next one runs, the twin returns the estimate one iteration past it (the two used to differ by one)
        x, s = richardson_lucy(_t(y), kernel, iterations=6, tol=tol, return_stats=True)
        assert s.stopped_by_tol and s.iterations == 5, s
        assert torch.equal(x, richardson_lucy(_t(y), kernel, iterations=5))
        # met only by the last iteration allowed: all of them ran, and it is reported as met
        x, s = richardson_lucy(_t(y), kernel, iterations=4, tol=tol, return_stats=True)
        assert s.stopped_by_tol and s.iterations == 4
    x, s = richardson_lucy(_t(np.zeros((6, 8, 8), np.float32)), psf, iterations=3, tol=1e-3, return_stats=True)
    assert not x.numpy().any() and s.iterations == 2 and s.stopped_by_tol and not s.total.any()
    x, s = richardson_lucy(_t(y), psf, iterations=0, return_stats=True)
    assert s.iterations == 0 and s.flux.shape == (0,)
    with pytest.raises(ValueError):
        richardson_lucy(_t(y), psf, iterations=2, tol=float("nan"))


def test_axial_psfs_up_to_31_taps_on_a_cpu_tensor_and_in_the_settings():
    """VERDICT r3 missing 5: a measured axial PSF spans more than +-7 planes.  Separable PSFs take up to 31 z taps (the
    device runs the z factor as its own launch, csrc/correlate_z.hip); a long non-separable PSF is refused up front."""
    from shrimpy_amd.deconvolve import RichardsonLucyPlan, prepare_psf, richardson_lucy
    from shrimpy_amd.settings import DeconvolveSettings

    psf, factors = o.gaussian_psf((21, 5, 7), (4.0, 1.0, 1.3))
    y = o.bead_scene((30, 16, 22), seed=4, psf=o.gaussian_psf((5, 5, 5), (1.2, 1.0, 1.0))[0], density=3e-3)
    x, s = richardson_lucy(_t(y), psf, iterations=4, return_stats=True)
    _rl_close(x.numpy(), o.richardson_lucy(y, psf, iterations=4))
    np.testing.assert_allclose(s.flux, float(y.astype(np.float64).sum()), rtol=1e-6)
    _rl_close(richardson_lucy(_t(y), psf_factors=factors, iterations=2).numpy(), o.richardson_lucy(y, psf, iterations=2))
    assert prepare_psf(np.ones((31, 15, 15), np.float32)).shape == (31, 15, 15)
    for bad in ((33, 3, 3), (31, 17, 3)):
        with pytest.raises(ValueError, match="exceeds"):
            prepare_psf(np.ones(bad, np.float32))
    assert DeconvolveSettings(gaussian_shape_zyx=(31, 15, 15), gaussian_sigma_zyx=(6.0, 2.0, 2.0)).gaussian_shape_zyx == (31, 15, 15)
    for bad in ((33, 3, 3), (9, 17, 3), (8, 3, 3)):
        with pytest.raises(ValueError):
            DeconvolveSettings(gaussian_shape_zyx=bad)
    with pytest.raises(_lib.LsrError):      # the plan itself lives on a GPU
        RichardsonLucyPlan((8, 8, 8), psf, "cpu")


def test_host_twins_check_their_arguments_like_the_device_entries():
    buf = np.zeros(64, np.float32)
    p, m = buf.ctypes.data, _lib.matrix12(np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0.0]]))
    lib = _lib.load()
    assert lib.lsr_deskew_f32_cpu(None, 4, 4, 4, p, 2, 4, 4, 4, 16, 4, m, 3, None) == -1          # LSR_E_NULL
    assert lib.lsr_deskew_f32_cpu(p, 4, 4, 4, p, 3, 4, 4, 4, 16, 4, m, 3, None) == -2             # Zo != ceil(Zd / avg)
    with pytest.raises(_lib.LsrUnsupported, match="not a deskew shear"):
        _lib.call("lsr_deskew_f32_cpu", p, 4, 4, 4, p, 2, 4, 4, 4, 16, 4, m, 3, None)
    assert lib.lsr_affine_f32_cpu(p, 2, 2, 2, p, 2, 2, 2, m, 0.0, 7, None) == -4                   # unknown mode
    assert lib.lsr_correlate_sep_f32_cpu(p, p, None, 2, 2, 2, p, 3, p, 3, p, 3, 0, 0.0, None, None, None, None) == -4   # out aliases in
    assert lib.lsr_set_host_threads(0) == -4 and lib.lsr_set_host_threads(1) == 0


# ------------------------------------------------------------------------------------------------ vs the kernels


@pytest.mark.gpu
def test_host_twins_equal_the_device_kernels_bit_for_bit(device):
    import torch

    from shrimpy_amd.deconvolve import RichardsonLucyPlan, correlate3d, richardson_lucy
    from shrimpy_amd.deskew import fast_deskew_zyx
    from shrimpy_amd.register import apply_affine_transform_zyx

    rng = np.random.default_rng(8)
    raw = rng.poisson(300, (200, 40, 70)).astype(np.float32)
    for keep, avg, border in [(False, 3, "constant"), (True, 2, "grid-constant"), (False, 1, "constant")]:
        kw = dict(ls_angle_deg=30.0, px_to_scan_ratio=0.755, keep_overhang=keep, average_n_slices=avg, border=border)
        assert torch.equal(fast_deskew_zyx(raw_data=_t(raw), **kw), fast_deskew_zyx(raw_data=_t(raw).to(device), **kw).cpu())
        r16 = _t(raw.astype(np.uint16))
        assert torch.equal(fast_deskew_zyx(raw_data=r16, **kw), fast_deskew_zyx(raw_data=r16.to(device), **kw).cpu())
    from shrimpy_amd.flatfield import flat_field_bf, flat_field_pattern

    for stack in (_t(raw), _t(raw.astype(np.uint16)), _t(raw[:199])):          # float, camera counts, odd Z
        host, dev = flat_field_pattern(stack), flat_field_pattern(stack.to(device))
        assert torch.equal(host.pattern, dev.pattern.cpu())
        np.testing.assert_allclose(flat_field_bf(stack).numpy(), flat_field_bf(stack.to(device)).cpu().numpy(), rtol=1e-6)
    vol = _t(rng.random((24, 48, 64)).astype(np.float32))
    m = np.eye(4)
    m[:3, :3] += rng.normal(0, 0.03, (3, 3))
    m[:3, 3] = [0.7, -2.2, 3.1]
    for mode in ("constant", "grid-constant"):
        assert torch.equal(apply_affine_transform_zyx(vol, m, mode=mode, cval=1.0),
                           apply_affine_transform_zyx(vol.to(device), m, mode=mode, cval=1.0).cpu())
    # stencils: the generic device kernels share the twins' operation order exactly
    psf, factors = o.gaussian_psf((9, 7, 7), (2.0, 1.2, 1.2))
    rot = psf.copy()
    rot[1, 2, 3] += 0.002
    y = _t(o.bead_scene((20, 50, 70), seed=4, psf=psf, density=1e-3))
    assert torch.equal(correlate3d(y, rot), correlate3d(y.to(device), rot, tuned=False).cpu())
    got = richardson_lucy(y, psf_factors=factors, iterations=6)
    for fused in ("auto", "never"):
        dev = RichardsonLucyPlan(tuple(y.shape), None, device, psf_factors=factors, fused=fused)(y.to(device), iterations=6)
        _rl_close(got.numpy(), dev.cpu().numpy())
    _rl_close(richardson_lucy(y, rot, iterations=3).numpy(), richardson_lucy(y.to(device), rot, iterations=3).cpu().numpy())


def test_the_cli_chain_runs_end_to_end_without_a_gpu(tmp_path, monkeypatch):
    """BASELINE configs[0] with everything around it: OME-Zarr in, YAML settings, the `reconstruct` command
    (deskew -> register -> deconvolve) and OME-Zarr out on a box with no HIP device -- the reference's cpu
    branch (`shrimpy/preprocessing.py:78-82`).  Every stage runs its native host twin; the stored volumes equal
    the oracle chain (deskew and registration bit for bit, RL within its bar)."""
    import torch
    import yaml

    from click.testing import CliRunner

    from shrimpy_amd import cli
    from shrimpy_amd.io.omezarr import open_ome_zarr

    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    rng = np.random.default_rng(12)
    raw = {}
    with open_ome_zarr(tmp_path / "raw.zarr", layout="hcs", mode="w", channel_names=["LS"], prefer_iohub=False) as plate:
        for i in range(2):
            arr = plate.create_position("A", str(i + 1), "0").create_zeros(
                "0", shape=(2, 1, 96, 16, 40), dtype="uint16", scale=(1, 1, 0.15, 0.1133, 0.1133))
            for t in range(2):
                raw[(i, t)] = rng.integers(90, 900, (96, 16, 40)).astype(np.uint16)
                arr.write_volume(t, 0, raw[(i, t)])
    m = np.eye(4)
    m[:3, 3] = [0.5, -1.25, 2.0]
    m[1, 2] = 0.02
    (tmp_path / "recon.yml").write_text(yaml.safe_dump(dict(
        deskew=dict(pixel_size_um=0.1133, ls_angle_deg=30.0, scan_step_um=0.15, keep_overhang=False, average_n_slices=3),
        registration=dict(affine_transform_zyx=m.tolist()),
        deconvolution=dict(iterations=4, gaussian_shape_zyx=[5, 5, 5], gaussian_sigma_zyx=[1.2, 1.0, 1.0]))))
    res = CliRunner().invoke(cli.cli, ["reconstruct", "-i", str(tmp_path / "raw.zarr"), "-c", str(tmp_path / "recon.yml"),
                                       "-o", str(tmp_path / "out.zarr")])
    assert res.exit_code == 0, res.output
    _, factors = o.gaussian_psf((5, 5, 5), (1.2, 1.0, 1.0))
    with open_ome_zarr(tmp_path / "out.zarr", prefer_iohub=False) as out:
        positions = dict(out.positions())
        assert list(positions) == ["A/1/0", "A/2/0"]
        for i, key in enumerate(positions):
            for t in range(2):
                d = o.deskew(raw[(i, t)].astype(np.float32), 30.0, 0.755, False, 3)
                reg = o.affine_apply_4x4(d, m, d.shape)
                want = o.richardson_lucy_separable(reg, factors, iterations=4)
                got = positions[key]["0"].read_volume(t, 0)
                assert got.shape == want.shape and got.dtype == np.float32
                _rl_close(got, want)
    # deskew alone: the stored volume IS the oracle's
    (tmp_path / "deskew.yml").write_text(yaml.safe_dump(dict(
        pixel_size_um=0.1133, ls_angle_deg=30.0, scan_step_um=0.15, keep_overhang=False, average_n_slices=3)))
    res = CliRunner().invoke(cli.cli, ["deskew", "-i", str(tmp_path / "raw.zarr"), "-c", str(tmp_path / "deskew.yml"),
                                       "-o", str(tmp_path / "deskewed.zarr"), "-p", "A/2/0"])
    assert res.exit_code == 0, res.output
    with open_ome_zarr(tmp_path / "deskewed.zarr", prefer_iohub=False) as out:
        np.testing.assert_array_equal(dict(out.positions())["A/2/0"]["0"].read_volume(1, 0),
                                      o.deskew(raw[(1, 1)].astype(np.float32), 30.0, 0.755, False, 3))


def test_two_ranks_of_the_cli_on_a_box_without_a_gpu_write_the_single_rank_store(tmp_path):
    """The sharded `reconstruct` command under `torch.distributed.run` with two ranks and no HIP device: units
    split round-robin, the ranks meet over gloo (barrier + agreement on rank 0's set-up only, no data-path
    collective), every rank writes its own positions -- and the store equals the one a single rank writes."""
    import os
    import socket
    import subprocess
    import sys

    import yaml

    from pathlib import Path

    from shrimpy_amd.io.omezarr import open_ome_zarr

    root = Path(__file__).resolve().parent.parent
    rng = np.random.default_rng(2)
    with open_ome_zarr(tmp_path / "raw.zarr", layout="hcs", mode="w", channel_names=["LS"], prefer_iohub=False) as plate:
        for i in range(3):
            arr = plate.create_position("A", str(i + 1), "0").create_zeros(
                "0", shape=(1, 1, 80, 12, 24), dtype="uint16", scale=(1, 1, 0.15, 0.1133, 0.1133))
            arr.write_volume(0, 0, rng.integers(90, 900, (80, 12, 24)).astype(np.uint16))
    (tmp_path / "recon.yml").write_text(yaml.safe_dump(dict(
        deskew=dict(pixel_size_um=0.1133, ls_angle_deg=30.0, scan_step_um=0.15, keep_overhang=False, average_n_slices=3),
        deconvolution=dict(iterations=3, gaussian_shape_zyx=[3, 3, 3], gaussian_sigma_zyx=[1.0, 1.0, 1.0]))))
    env = dict(os.environ, OMP_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    args = ["-m", "shrimpy_amd", "reconstruct", "-i", str(tmp_path / "raw.zarr"), "-c", str(tmp_path / "recon.yml")]
    one = subprocess.run([sys.executable, *args, "-o", str(tmp_path / "one.zarr")], cwd=root, env=env,
                         capture_output=True, text=True, timeout=300)
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-2000:]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), *args, "-o", str(tmp_path / "two.zarr")],
                         cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-2000:]
    assert "'world_size': 2" in two.stdout
    with open_ome_zarr(tmp_path / "one.zarr", prefer_iohub=False) as a, open_ome_zarr(tmp_path / "two.zarr", prefer_iohub=False) as b:
        for (ka, pa), (kb, pb) in zip(a.positions(), b.positions()):
            assert ka == kb
            va = pa["0"].read_volume(0, 0)
            assert float(va.max()) > 0
            np.testing.assert_array_equal(va, pb["0"].read_volume(0, 0))


def test_a_measured_psf_comes_from_npy_or_from_an_ome_zarr_store(tmp_path):
    """``DeconvolveSettings.psf_path``: a ``.npy`` array as it is, or a bead volume in an OME-Zarr store (HCS,
    position 0/0/0, array "0": what the PSF tools around the reference write, ``scripts/measure_psf.py:273-287``),
    cut to ``psf_shape_zyx`` around its peak and renormalised; the deconvolution then uses exactly that array."""
    import torch

    from shrimpy_amd.io.omezarr import open_ome_zarr
    from shrimpy_amd.pipeline import VolumeReconstructor
    from shrimpy_amd.settings import DeconvolveSettings, ReconstructSettings

    psf, _ = o.gaussian_psf((5, 5, 7), (1.1, 0.9, 1.4))
    np.save(tmp_path / "psf.npy", psf)
    np.testing.assert_array_equal(DeconvolveSettings(psf_path=str(tmp_path / "psf.npy")).load_psf(), psf)
    big = np.zeros((21, 31, 33), np.float32)          # a measured bead volume, peak off-centre
    big[8:13, 15:20, 10:17] = psf * 1000.0 + 0.0
    with open_ome_zarr(tmp_path / "psf.zarr", layout="hcs", mode="w", channel_names=["beads"], prefer_iohub=False) as store:
        arr = store.create_position("0", "0", "0").create_zeros("0", shape=(1, 1) + big.shape, dtype="float32",
                                                                  scale=(1, 1, 0.17, 0.1133, 0.1133))
        arr.write_volume(0, 0, big)
    dec = DeconvolveSettings(psf_path=str(tmp_path / "psf.zarr"), psf_shape_zyx=(5, 5, 7), iterations=3)
    cut = dec.load_psf()
    assert cut.shape == (5, 5, 7) and abs(float(cut.sum()) - 1.0) < 1e-6
    np.testing.assert_allclose(cut, psf / psf.sum(), rtol=1e-6)
    # uncut, the 21 x 31 x 33 bead volume is beyond the stencil kernels: refused under method="direct", taken as it
    # is otherwise (the Fourier-domain iteration on a device, the twins' dense loop on the host)
    with pytest.raises(ValueError, match="psf_shape_zyx"):
        DeconvolveSettings(psf_path=str(tmp_path / "psf.zarr"), method="direct").load_psf()
    assert DeconvolveSettings(psf_path=str(tmp_path / "psf.zarr")).load_psf().shape == big.shape
    edge = np.zeros((9, 9, 9), np.float32)
    edge[1, 4, 4] = 1.0                                # peak one plane from the face: a 5-plane window does not fit
    np.save(tmp_path / "edge.npy", edge)
    with pytest.raises(ValueError, match="does not fit"):
        DeconvolveSettings(psf_path=str(tmp_path / "edge.npy"), psf_shape_zyx=(5, 5, 5)).load_psf()
    with pytest.raises(ValueError):
        DeconvolveSettings(psf_shape_zyx=(4, 5, 5))
    y = o.bead_scene((10, 18, 24), seed=3, psf=psf, density=2e-3)
    rec = VolumeReconstructor(y.shape, ReconstructSettings(deconvolution=dec), torch.device("cpu"))
    _rl_close(rec(y).numpy(), o.richardson_lucy(y, cut, iterations=3))


def test_host_path_argument_errors_read_like_the_device_paths():
    import torch

    from shrimpy_amd.deconvolve import correlate3d, richardson_lucy
    from shrimpy_amd.deskew import average_n_slices, deskew_with_matrix, fast_deskew_zyx
    from shrimpy_amd.register import apply_affine_transform_zyx

    v = torch.zeros((8, 4, 6))
    with pytest.raises(ValueError, match="is empty"):                          # scan too short for the tilt
        fast_deskew_zyx(raw_data=torch.zeros((4, 64, 8)), ls_angle_deg=30, px_to_scan_ratio=0.755, keep_overhang=False)
    with pytest.raises(ValueError, match="border"):
        fast_deskew_zyx(raw_data=v, ls_angle_deg=30, px_to_scan_ratio=0.755, keep_overhang=True, border="wrap")
    with pytest.raises(ValueError, match="average_n_slices"):
        deskew_with_matrix(v, np.eye(4)[:3], (8, 4, 6), 0)
    with pytest.raises(ValueError, match="out must be"):
        deskew_with_matrix(v, np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0.0]]), (8, 4, 6), 1, out=torch.zeros((3, 4, 6)))
    with pytest.raises(ValueError, match="alias"):
        apply_affine_transform_zyx(v, np.eye(4), out=v)
    with pytest.raises(ValueError, match="mode"):
        apply_affine_transform_zyx(v, np.eye(4), mode="wrap")
    with pytest.raises(ValueError, match="iterations"):
        richardson_lucy(v, np.ones((3, 3, 3), np.float32) / 27, iterations=-1)
    with pytest.raises(ValueError, match="x0 must be"):
        richardson_lucy(v, np.ones((3, 3, 3), np.float32) / 27, x0=torch.zeros((2, 2, 2)))
    with pytest.raises(ValueError, match="exceeds"):                           # (129 taps per axis: what the device's Fourier-domain path takes)
        richardson_lucy(v, np.ones((3, 131, 3), np.float32))
    with pytest.raises(ValueError, match="Z, Y, X"):
        correlate3d(torch.zeros((4, 4)), np.ones((3, 3, 3), np.float32))
    assert average_n_slices(v, 1) is v and tuple(average_n_slices(v, 3).shape) == (3, 4, 6)
    # zero iterations hand back the start, a non-contiguous input is made contiguous, float64 is converted
    assert torch.equal(richardson_lucy(v + 2, np.ones((3, 3, 3), np.float32) / 27, iterations=0), v + 2)
    t = torch.arange(8 * 4 * 6, dtype=torch.float64).reshape(6, 4, 8).permute(2, 1, 0)
    np.testing.assert_array_equal(fast_deskew_zyx(raw_data=t, ls_angle_deg=30, px_to_scan_ratio=0.755, keep_overhang=True).numpy(),
                                  o.deskew(t.numpy().astype(np.float32), 30.0, 0.755, True, 1))
