"""CPU tests of the host side: geometry, settings, C-ABI surface, loud failure without a GPU.

No compute call is made here (there is no GPU in the build container).
"""

import ctypes
import inspect
import re

import numpy as np
import pytest

from oracle import cpu_ref as o
from shrimpy_amd import _lib, geometry
from shrimpy_amd.settings import (
    DeconvolveSettings,
    DeskewSettings,
    ReconstructSettings,
    RegisterSettings,
)

# ---------------------------------------------------------------- geometry vs the oracle's own


@pytest.mark.parametrize("raw_shape", [(24, 16, 40), (2048, 512, 2048), (256, 64, 256), (1201, 256, 1600)])
@pytest.mark.parametrize("keep_overhang", [False, True])
@pytest.mark.parametrize("avg", [1, 3, 5])
def test_geometry_agrees_with_oracle(raw_shape, keep_overhang, avg):
    geo = geometry.deskew_geometry(raw_shape, 30.0, 0.755, keep_overhang, avg, 0.1133)
    m, off, pre = o.deskew_geometry(raw_shape, 30.0, 0.755, keep_overhang)
    np.testing.assert_array_equal(geo.matrix_3x4[:, :3], m)
    np.testing.assert_array_equal(geo.matrix_3x4[:, 3], off)
    assert geo.pre_average_shape == pre
    shape, voxel = o.deskewed_shape(raw_shape, 30.0, 0.755, keep_overhang, avg, 0.1133)
    assert geo.output_shape == shape
    assert geo.voxel_size == pytest.approx(voxel)


def test_get_deskewed_data_shape_signature_is_the_reference_call():
    """``shrimpy/preprocessing.py:226-231`` calls it with ``raw_data_shape=`` + filtered settings;
    ``scripts/measure_psf.py:230-234`` adds ``pixel_size_um=``."""
    from shrimpy_amd.deskew import fast_deskew_zyx, get_deskewed_data_shape

    params = list(inspect.signature(get_deskewed_data_shape).parameters)
    # the reference's parameters first, in its order; the convention switches after them, defaulted
    assert params == ["raw_data_shape", "ls_angle_deg", "px_to_scan_ratio", "keep_overhang",
                      "average_n_slices", "pixel_size_um", "orientation"]
    params = list(inspect.signature(fast_deskew_zyx).parameters)
    assert params == ["raw_data", "ls_angle_deg", "px_to_scan_ratio", "keep_overhang", "average_n_slices",
                      "orientation", "border", "cval"]
    sig = inspect.signature(fast_deskew_zyx).parameters
    assert sig["orientation"].default == "identity" and sig["border"].default == "constant" and sig["cval"].default == 0.0
    shape, voxel = get_deskewed_data_shape(
        raw_data_shape=(2048, 512, 2048), ls_angle_deg=30, px_to_scan_ratio=0.755,
        keep_overhang=False, average_n_slices=3, pixel_size_um=0.1133)
    assert shape == (171, 2048, 2270)
    assert len(voxel) == 3


def test_empty_deskew_is_reported_by_the_shape_rule():
    # tilt longer than the scan: the no-overhang window is empty (SURVEY section 8 preamble)
    shape, _ = geometry.deskew_geometry((512, 2048, 64), 30.0, 0.755, False)[2:], None
    assert shape[0][2] < 0


def test_geometry_rejects_bad_input():
    with pytest.raises(ValueError):
        geometry.deskew_geometry((4, 4), 30, 0.755, False)
    with pytest.raises(ValueError):
        geometry.deskew_geometry((4, 0, 4), 30, 0.755, False)
    with pytest.raises(ValueError):
        geometry.deskew_geometry((4, 4, 4), 30, 0.0, False)
    with pytest.raises(ValueError):
        geometry.deskew_geometry((4, 4, 4), 30, 0.755, False, average_n_slices=0)
    with pytest.raises(ValueError):
        geometry.as_matrix_3x4(np.eye(3))
    with pytest.raises(ValueError):
        geometry.as_matrix_3x4(np.full((3, 4), np.nan))


# ---------------------------------------------------------------- settings


def test_deskew_settings_derive_ratio_like_the_reference_scripts():
    """``scripts/measure_psf.py:225``: ratio = round(pixel / scan_step, 3); scale injection
    supplies pixel_size_um + scan_step_um (``shrimpy/dynatrack/manager.py:297-299``)."""
    s = DeskewSettings(ls_angle_deg=30.0, keep_overhang=False, average_n_slices=3,
                       pixel_size_um=0.1133, scan_step_um=0.15)
    assert s.px_to_scan_ratio == 0.755
    d = s.model_dump()
    assert set(d) == {"pixel_size_um", "ls_angle_deg", "px_to_scan_ratio", "scan_step_um",
                      "keep_overhang", "average_n_slices", "orientation", "border", "cval"}
    assert (d["orientation"], d["border"], d["cval"]) == ("identity", "constant", 0.0)
    # attributes the reference reads by getattr (shrimpy/preprocessing.py:240-242)
    assert (s.px_to_scan_ratio, s.pixel_size_um, s.scan_step_um) == (0.755, 0.1133, 0.15)


def test_deskew_settings_defaults_and_validation():
    s = DeskewSettings(pixel_size_um=0.1, ls_angle_deg=30, px_to_scan_ratio=0.75549)
    assert s.keep_overhang is False and s.average_n_slices == 3 and s.px_to_scan_ratio == 0.755
    with pytest.raises(ValueError):
        DeskewSettings(pixel_size_um=0.1, ls_angle_deg=30)  # neither ratio nor scan step
    with pytest.raises(ValueError):
        DeskewSettings(pixel_size_um=0.1, ls_angle_deg=50, px_to_scan_ratio=0.7)
    with pytest.raises(ValueError):
        DeskewSettings(pixel_size_um=0.1, ls_angle_deg=30, px_to_scan_ratio=0.7, bogus=1)  # extra=forbid
    with pytest.raises(ValueError):
        DeskewSettings(pixel_size_um=0.1, ls_angle_deg=30, px_to_scan_ratio=0.7, average_n_slices=0)


def test_settings_kwargs_filtering_keeps_exactly_the_callee_params(golden_dir):
    """The reference filters ``model_dump()`` by the callee signature
    (``shrimpy/preprocessing.py:44-56``); the kept set was captured from the reference itself."""
    from shrimpy_amd.deskew import fast_deskew_zyx

    from shrimpy_amd.preprocessing import accepted_kwargs

    s = DeskewSettings(ls_angle_deg=30.0, pixel_size_um=0.1133, scan_step_um=0.15)
    ref = list(np.load(golden_dir / "ref_preprocessing.npz")["settings_kwargs_kept"])

    class BiahubShaped:   # the six fields the reference's settings model carries (no switches)
        def model_dump(self):
            d = s.model_dump()
            return {k: d[k] for k in ("ls_angle_deg", "px_to_scan_ratio", "keep_overhang", "average_n_slices",
                                      "pixel_size_um", "scan_step_um")}

    assert sorted(accepted_kwargs(fast_deskew_zyx, BiahubShaped())) == ref
    # our own model adds exactly the three convention switches, and they do reach the callee
    assert sorted(accepted_kwargs(fast_deskew_zyx, s)) == sorted(ref + ["orientation", "border", "cval"])
    assert DeskewSettings(ls_angle_deg=30.0, pixel_size_um=0.1, px_to_scan_ratio=0.7, cval="min").cval == "min"
    assert DeskewSettings(ls_angle_deg=30.0, pixel_size_um=0.1, px_to_scan_ratio=0.7, cval=None).cval is None
    with pytest.raises(ValueError):
        DeskewSettings(ls_angle_deg=30.0, pixel_size_um=0.1, px_to_scan_ratio=0.7, cval="max")


def test_register_and_deconvolve_settings(tmp_path):
    m = np.eye(4).tolist()
    r = RegisterSettings(affine_transform_zyx=m)
    assert r.mode == "constant" and r.cval == 0.0
    with pytest.raises(ValueError):
        RegisterSettings(affine_transform_zyx=np.eye(3).tolist())
    with pytest.raises(ValueError):
        bad = np.eye(4)
        bad[3, 0] = 1
        RegisterSettings(affine_transform_zyx=bad.tolist())
    d = DeconvolveSettings()
    assert d.iterations == 20 and d.gaussian_shape_zyx == (9, 7, 7)
    with pytest.raises(ValueError):
        DeconvolveSettings(gaussian_shape_zyx=(8, 7, 7))
    full = ReconstructSettings(
        deskew=DeskewSettings(pixel_size_um=0.1133, ls_angle_deg=30, scan_step_um=0.15),
        registration=r, deconvolution=d)
    p = tmp_path / "recon.yml"
    full.to_yaml(p)
    again = ReconstructSettings.from_yaml(p)
    assert again == full


# ---------------------------------------------------------------- C ABI surface


def _declared_symbols():
    text = _lib.HEADER_PATH.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lsr_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_symbol_the_header_declares():
    assert _lib.LIB_PATH.exists(), "build liblsrecon.so first (__graft_entry__.build())"
    lib = ctypes.CDLL(str(_lib.LIB_PATH))
    declared = _declared_symbols()
    assert "lsr_deskew_f32" in declared and "lsr_rl_sep_f32" in declared
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/lsrecon.h but not exported"
    # and the ctypes table binds exactly the declared compute entry points
    assert sorted(set(_lib.SIGNATURES) | {"lsr_last_error", "lsr_source_sha16"}) == declared


def test_the_binary_carries_the_stamp_of_the_sources_beside_it():
    assert _lib.library_source_sha16() == _lib.kernel_source_sha16()
    assert re.fullmatch(r"[0-9a-f]{16}", _lib.library_source_sha16())


def test_a_stale_library_is_refused_at_load_time(tmp_path):
    """the .so travels as a file; a build of OTHER sources must not load silently.  The stale
    build here is the real library with the stamp bytes of its read-only data changed."""
    import os
    import subprocess
    import sys

    blob = _lib.LIB_PATH.read_bytes()
    stamp = _lib.kernel_source_sha16().encode()
    assert blob.count(stamp) == 1, "the stamp string appears once in the binary"
    stale = tmp_path / "liblsrecon.so"
    stale.write_bytes(blob.replace(stamp, b"0123456789abcdef"))
    code = ("from shrimpy_amd import _lib\n"
            "try:\n    _lib.load()\n    print('LOADED')\n"
            "except _lib.LsrError as e:\n    print('REFUSED', e)\n")
    env = dict(os.environ, LSR_LIBRARY=str(stale))
    env.pop("LSR_ALLOW_STALE_LIBRARY", None)
    root = str(_lib.HEADER_PATH.parent.parent)
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert out.stdout.startswith("REFUSED"), out.stdout + out.stderr
    assert "0123456789abcdef" in out.stdout and stamp.decode() in out.stdout and "rebuild" in out.stdout
    # the override loads it, loudly
    out = subprocess.run([sys.executable, "-W", "always", "-c", code], env=dict(env, LSR_ALLOW_STALE_LIBRARY="1"), cwd=root,
                         capture_output=True, text=True, timeout=300)
    assert out.stdout.startswith("LOADED") and "stale liblsrecon.so" in out.stderr, out.stdout + out.stderr


def test_library_loads_and_reports_version_and_arg_errors():
    lib = _lib.load()
    assert lib.lsr_version() == 100
    # argument validation happens before any launch: safe without a GPU
    rc = lib.lsr_deskew_f32(None, 1, 1, 1, None, 1, 1, 1, 1, 1, 1, None, 1, None)
    assert rc == -1 and b"NULL" in lib.lsr_last_error()
    m = _lib.matrix12(np.eye(3, 4))
    buf = ctypes.create_string_buffer(64)
    p = ctypes.cast(buf, ctypes.c_void_p)
    rc = lib.lsr_deskew_f32(p, 4, 4, 4, p, 2, 4, 4, 4, 16, 4, m, 3, None)
    assert rc == -3, "identity is not a deskew shear -> LSR_E_UNSUPPORTED"
    rc = lib.lsr_deskew_f32(p, 4, 0, 4, p, 2, 4, 4, 4, 16, 4, m, 3, None)
    assert rc == -2
    rc = lib.lsr_correlate_sep_f32(p, p, None, 4, 4, 4, p, 4, p, 3, p, 3, 0, 0.0, None, None, None, None)
    assert rc == -3, "even tap count is unsupported"
    rc = lib.lsr_affine_f32(p, 2, 2, 2, p, 2, 2, 2, m, 0.0, 7, None)
    assert rc == -4
    with pytest.raises(_lib.LsrUnsupported):
        _lib.call("lsr_deskew_f32", p, 4, 4, 4, p, 2, 4, 4, 4, 16, 4, m, 3, None)


def test_gpu_only_objects_fail_loudly_on_cpu():
    """CPU tensors run the native host twins (tests/test_host_twins.py); what exists only on a HIP device --
    the staged store path, the RL plan with its padded device volumes -- says so."""
    from shrimpy_amd.deconvolve import RichardsonLucyPlan
    from shrimpy_amd.staging import VolumeStager

    with pytest.raises(_lib.LsrError, match="not a GPU.*host twins"):
        RichardsonLucyPlan((8, 8, 8), np.ones((3, 3, 3), np.float32) / 27, "cpu")
    with pytest.raises(_lib.LsrError, match="needs a HIP device"):
        VolumeStager((8, 8, 8), "uint16", (4, 8, 8), "cpu")


def test_product_never_imports_the_oracle():
    import pathlib

    pkg = pathlib.Path(_lib.__file__).parent
    for f in pkg.rglob("*.py"):
        src = f.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
        assert not re.search(r"^\s*(from|import)\s+scipy\b", src, flags=re.M), (
            f"{f} must not import scipy (the CPU reference path)")


# ---------------------------------------------------------------- PSF preparation (host math)


def test_factor_psf_recovers_separable_factors_and_rejects_rotated():
    from shrimpy_amd.deconvolve import factor_psf, prepare_psf

    psf, (kz, ky, kx) = o.gaussian_psf()
    f = factor_psf(prepare_psf(psf))
    assert f is not None
    rec = f[0][:, None, None] * f[1][None, :, None] * f[2][None, None, :]
    np.testing.assert_allclose(rec, psf, atol=1e-6 * psf.max())
    assert factor_psf(prepare_psf(o.rotated_psf())) is None


def test_axis_norm_and_prefix_table_match_oracle_norm():
    from shrimpy_amd.deconvolve import _axis_norm, _prefix_table

    psf, (kz, ky, kx) = o.gaussian_psf((5, 3, 7), (1.0, 0.8, 1.5))
    shape = (6, 9, 11)
    ref = o.rl_norm(shape, psf)
    sep = (_axis_norm(kz, shape[0])[:, None, None] * _axis_norm(ky, shape[1])[None, :, None]
           * _axis_norm(kx, shape[2])[None, None, :])
    np.testing.assert_allclose(sep, ref, rtol=1e-6)
    t = _prefix_table(psf)
    assert t.shape == (6, 4, 8) and t[0].max() == 0 and abs(t[-1, -1, -1] - psf.sum(dtype=np.float64)) < 1e-12
    # volume narrower than the PSF along z: both borders clip at once
    thin = o.rl_norm((3, 9, 11), psf)
    sep = (_axis_norm(kz, 3)[:, None, None] * _axis_norm(ky, 9)[None, :, None]
           * _axis_norm(kx, 11)[None, None, :])
    np.testing.assert_allclose(sep, thin, rtol=1e-6)


def test_prepare_psf_pads_even_axes_and_bounds_size():
    from shrimpy_amd.deconvolve import prepare_psf

    assert prepare_psf(np.ones((2, 4, 3), np.float32)).shape == (3, 5, 3)
    assert prepare_psf(np.ones((17, 3, 3), np.float32)).shape == (17, 3, 3)     # long z: separable PSFs (round 4)
    for bad in ((33, 3, 3), (3, 17, 3), (3, 3, 17)):
        with pytest.raises(ValueError):
            prepare_psf(np.ones(bad, np.float32))
    with pytest.raises(ValueError):
        prepare_psf(np.ones((3, 3), np.float32))


def test_dynatrack_host_logic_matches_the_oracle(golden_dir):
    """The 256-bin searches of shrimpy_amd.dynatrack run on the host: check them without a GPU
    (histograms from the oracle) against the oracle, which is pinned to the reference's output."""
    import sys

    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    from oracle import cpu_ref as o
    from shrimpy_amd import dynatrack as d

    g = np.load(golden_dir / "ref_dynatrack.npz")
    a = g["a"]
    blur = o.dt_gaussian_blur_3d((a - a.min()) / (a.max() - a.min()), 2.0)
    vmin, vmax = float(blur.min()), float(blur.max())
    hist = o.dt_histc(blur, 256, vmin, vmax)
    for comp, want in zip((0, 1), g["otsu_blur_a"]):
        assert d._otsu_from_hist(hist, vmin, vmax, comp) == pytest.approx(float(want), rel=1e-5)
    assert [d._next_fast_len(n) for n in (0, 1, 7, 11, 13, 171, 2049)] == [1, 1, 8, 12, 15, 180, 2160]


# ---------------------------------------------------------------- orientation / border switches

ORIENTATIONS = ["identity", "flip_z", "flip_y", "flip_x", "transpose_yx", "rot90", "rot180", "rot270",
                "flip_z+rot90", "rot90+flip_y", "transpose_yx+flip_x+flip_z"]


@pytest.mark.parametrize("spec", ORIENTATIONS)
def test_orient_axes_is_what_numpy_does(spec):
    """``geometry.orient_axes`` (what the device post-step applies as permute + flip) against the
    oracle's literal numpy flips / rot90 on an index volume."""
    shape = (3, 4, 5)
    vol = np.arange(np.prod(shape), dtype=np.float32).reshape(shape)
    perm, rev = geometry.orient_axes(spec)
    mine = np.transpose(vol, perm)
    for axis, r in enumerate(rev):
        if r:
            mine = np.flip(mine, axis)
    want = o.orient(vol, spec)
    np.testing.assert_array_equal(mine, want)
    assert geometry.orient_shape(shape, spec) == want.shape
    assert geometry.orient_voxel((0.17, 0.11, 0.12), spec) == tuple((0.17, 0.11, 0.12)[a] for a in perm)


def test_orientation_specs_are_validated():
    with pytest.raises(ValueError):
        geometry.parse_orientation("rot45")
    with pytest.raises(TypeError):
        geometry.parse_orientation(3)
    assert geometry.parse_orientation("identity+flip_z") == ("flip_z",)
    with pytest.raises(ValueError):
        DeskewSettings(pixel_size_um=0.1, ls_angle_deg=30, px_to_scan_ratio=0.7, orientation="upside-down")
    with pytest.raises(ValueError):
        DeskewSettings(pixel_size_um=0.1, ls_angle_deg=30, px_to_scan_ratio=0.7, border="reflect")
    s = DeskewSettings(pixel_size_um=0.1, ls_angle_deg=30, px_to_scan_ratio=0.7, orientation="rot90",
                       border="grid-constant")
    assert (s.orientation, s.border) == ("rot90", "grid-constant")


@pytest.mark.parametrize("spec", ORIENTATIONS)
def test_get_deskewed_data_shape_follows_the_orientation(spec):
    from shrimpy_amd.deskew import get_deskewed_data_shape

    base, voxel = get_deskewed_data_shape((40, 12, 16), 30.0, 0.755, False, 3, 0.1133)
    shape, vox = get_deskewed_data_shape((40, 12, 16), 30.0, 0.755, False, 3, 0.1133, orientation=spec)
    want = o.orient(np.zeros(base, np.float32), spec).shape
    assert shape == want
    assert sorted(vox) == sorted(voxel)


@pytest.mark.parametrize("border", ["constant", "grid-constant"])
@pytest.mark.parametrize("spec", ORIENTATIONS)
def test_chunk_reversal_identity_under_every_orientation(spec, border):
    """``scripts/measure_psf.py:221-249``: raw-X chunks are deskewed on their own and joined reversed
    on output axis -2.  With an orientation the join axis / order is ``raw_x_chunk_layout``; the
    identity must hold for the oracle under every switch (it is what lets the reference chunk)."""
    rng = np.random.default_rng(5)
    raw = rng.integers(80, 600, (40, 12, 16)).astype(np.float32)
    whole = o.deskew(raw, 30.0, 0.755, False, 3, orientation=spec, border=border)
    parts = [o.deskew(raw[:, :, a:a + 4], 30.0, 0.755, False, 3, orientation=spec, border=border)
             for a in range(0, 16, 4)]
    axis, reverse = geometry.raw_x_chunk_layout(spec)
    joined = np.concatenate(parts[::-1] if reverse else parts, axis=axis)
    np.testing.assert_array_equal(joined, whole)
    if spec == "identity":
        assert (axis, reverse) == (1, True)      # the reference's own join: reversed on axis -2


def test_border_switch_changes_only_the_border_band():
    """``grid-constant`` blends towards zero across the scan ends; the interior is the same sample."""
    rng = np.random.default_rng(6)
    raw = rng.integers(80, 600, (40, 12, 16)).astype(np.float32)
    a = o.deskew(raw, 30.0, 0.755, True, 1, border="constant")
    b = o.deskew(raw, 30.0, 0.755, True, 1, border="grid-constant")
    assert a.shape == b.shape and not np.array_equal(a, b)
    both = (a != 0) & (b != 0)
    np.testing.assert_allclose(a[both], b[both], rtol=1e-6)
    assert np.count_nonzero(b) > np.count_nonzero(a)   # blended samples exist only with grid-constant


def test_deskew_with_matrix_rejects_an_unknown_border():
    import torch

    from shrimpy_amd.deskew import deskew_with_matrix

    with pytest.raises(ValueError, match="border"):
        deskew_with_matrix(torch.zeros(4, 4, 4), np.zeros((3, 4)), (4, 4, 4), border="wrap")


# ---------------------------------------------------------------- host-side index math of the kernels
# (the entry points below run on the host; no GPU is touched)


def _box_shape(shape, m):
    import ctypes

    from shrimpy_amd import _lib

    out = (ctypes.c_int * 6)()
    ok = _lib.call_value("lsr_affine_box_shape", shape[0], shape[1], shape[2], _lib.matrix12(geometry.as_matrix_3x4(m)), out)
    return tuple(out) if ok else None


def _random_affine(rng, tilt=0.25):
    th = np.deg2rad(rng.uniform(-20, 20))
    m = np.eye(4)
    m[:3, :3] = np.array([[1, 0, 0], [0, np.cos(th), -np.sin(th)], [0, np.sin(th), np.cos(th)]]) @ np.diag(
        rng.uniform(0.6, 1.4, 3) * rng.choice([1.0, 1.0, -1.0], 3))
    m[0, 1:3] = rng.uniform(-tilt, tilt, 2)
    m[1:3, 0] = rng.uniform(-tilt, tilt, 2)
    m[:3, 3] = rng.uniform(-20, 200, 3)
    return m


def test_affine_box_covers_every_tap_of_every_block():
    """The box kernel stages, per block of output voxels, the source box spanned from the block's two
    extreme corners (no slack: the coordinate expression is monotone in each index).  Restated in numpy
    float64 in scipy's operation order: for random matrices and random block positions, every voxel's
    lower tap and its upper neighbour lie inside the box the library sizes."""
    rng = np.random.default_rng(12)
    shape = (300, 400, 512)
    checked = 0
    for _ in range(200):
        m = _random_affine(rng)
        got = _box_shape(shape, m)
        if got is None:
            continue
        tz, ty, tx, bz, by, bx = got
        assert tz * ty * tx == 4096 and bx % 4 == 0
        z0, y0, x0 = (int(rng.integers(0, 30)) * tz, int(rng.integers(0, 20)) * ty, int(rng.integers(0, 12)) * tx)
        zo, yo, xo = np.meshgrid(np.arange(z0, z0 + tz, dtype=np.float64), np.arange(y0, y0 + ty, dtype=np.float64),
                                 np.arange(x0, x0 + tx, dtype=np.float64), indexing="ij")
        for axis, n_box in enumerate((bz, by, bx)):
            r = m[axis]
            c = ((zo * r[0] + yo * r[1]) + xo * r[2]) + r[3]          # scipy's order, every step rounded
            lo = [z0 if r[0] >= 0 else z0 + tz - 1, y0 if r[1] >= 0 else y0 + ty - 1, x0 if r[2] >= 0 else x0 + tx - 1]
            hi = [z0 + tz - 1 if r[0] >= 0 else z0, y0 + ty - 1 if r[1] >= 0 else y0, x0 + tx - 1 if r[2] >= 0 else x0]
            cmin = ((lo[0] * r[0] + lo[1] * r[1]) + lo[2] * r[2]) + r[3]
            cmax = ((hi[0] * r[0] + hi[1] * r[1]) + hi[2] * r[2]) + r[3]
            assert cmin == c.min() and cmax == c.max()               # the corners ARE the extremes, exactly
            origin = np.floor(cmin) - (np.floor(cmin) % 4 if axis == 2 else 0)   # x origin: 16-byte aligned
            assert np.floor(c).max() + 1 - origin <= n_box - 1
        checked += 1
    assert checked > 100


def test_affine_path_is_decided_on_the_host():
    from shrimpy_amd import _lib

    def path(shape, m, mode=_lib.MODE_CONSTANT):
        return _lib.call_value("lsr_affine_path", shape[0], shape[1], shape[2], _lib.matrix12(geometry.as_matrix_3x4(m)), mode)

    ident = np.eye(4)
    tilt = np.eye(4)
    tilt[0, 2], tilt[2, 0] = -0.05, 0.05
    assert path((64, 256, 256), ident) == 1 and path((64, 256, 256), tilt) == 2
    assert path((64, 256, 254), tilt) == 0 and path((64, 256, 256), tilt, _lib.MODE_GRID_CONSTANT) == 2
    assert path((64, 256, 256), ident, _lib.MODE_GRID_CONSTANT) == 1      # both LDS-staged kernels: either border rule
    big = np.diag([9.0, 9.0, 9.0, 1.0])
    big[0, 2] = 0.1
    assert path((64, 256, 256), big) == 0 and _box_shape((64, 256, 256), big) is None
    # a tilt about y wants a short block along x, a tilt about x a short block along y
    about_y, about_x = np.eye(4), np.eye(4)
    about_y[0, 2], about_x[0, 1] = 0.3, 0.3
    sy, sx = _box_shape((64, 256, 256), about_y), _box_shape((64, 256, 256), about_x)
    assert sy[2] <= sx[2] and sx[1] <= sy[1]


def test_padded_shape_and_fused_support_tables():
    """``lsr_sep_padded_shape``: logical origin at row 2C / column 32, pitch a multiple of 32 floats,
    rows a whole number of tiles plus 4C; ``lsr_rl_sep_fused_supported``: odd tap counts up to 15, the
    spill-prone corner excluded."""
    import ctypes

    from shrimpy_amd import _lib

    for (y, x, pz, py, px) in [(2048, 2270, 9, 7, 7), (100, 70, 3, 3, 3), (1, 1, 15, 15, 15), (33, 129, 5, 9, 3)]:
        out = (ctypes.c_int64 * 4)()
        _lib.call("lsr_sep_padded_shape", y, x, pz, py, px, out)
        rows, pitch = out[0], out[1]
        c = max(py, px) // 2
        assert pitch % 32 == 0 and pitch >= x + 32 + 2 * c and rows >= y + 4 * c
    assert tuple(_lib.call_value("lsr_sep_padded_shape", 2048, 2270, 9, 7, 7, (ctypes.c_int64 * 4)()) for _ in range(1)) == (0,)
    ok = {(pz, pyx): _lib.call_value("lsr_rl_sep_fused_supported", pz, pyx, pyx) for pz in range(1, 17) for pyx in range(1, 17)}
    assert ok[(9, 7)] == 1 and ok[(3, 3)] == 1 and ok[(15, 9)] == 1
    assert ok[(15, 11)] == 0 and ok[(15, 15)] == 0          # would spill the accumulators
    assert all(v == 0 for (pz, pyx), v in ok.items() if pz % 2 == 0 or pyx % 2 == 0 or pz > 15 or pyx > 15)


def test_limit_shifts_cases_of_the_reference():
    """``TestLimitShiftsZyx`` (``shrimpy/tests/test_dynatrack.py:119-142``): deadband below the minimum, clip above
    the maximum with the sign kept, unchanged inside, axes without limits untouched -- host logic, no device."""
    from shrimpy_amd.dynatrack import _limit_shifts_zyx

    wide = {"z": (0.1, 10.0), "y": (0.1, 10.0), "x": (0.1, 10.0)}
    cases = [([0.5, 0.3, 0.1], {"z": (1.0, 10.0), "y": (1.0, 10.0), "x": (1.0, 10.0)}, [0.0, 0.0, 0.0]),
             ([15.0, -12.0, 8.0], wide, [10.0, -10.0, 8.0]),
             ([5.0, -3.0, 2.0], wide, [5.0, -3.0, 2.0]),
             ([5.0, 0.01, 2.0], {"z": (0.1, 10.0)}, [5.0, 0.01, 2.0])]
    for shifts, limits, want in cases:
        np.testing.assert_array_equal(_limit_shifts_zyx(np.array(shifts), limits), want)


def test_absurd_sizes_are_refused_before_anything_is_computed_with_them():
    """Extents of 2**30 and more, volumes of 2**48 voxels and more, strides past 2**31 / 2**32: LSR_E_UNSUPPORTED from
    every entry, device or host -- the planning arithmetic behind the checks never sees them (found by
    ``tools/fuzz_device_args.py`` under UBSan: signed overflows and a division by zero in tile planning)."""
    import ctypes

    from shrimpy_amd import _lib

    lib = _lib.load()
    buf = np.zeros(4096, np.float32)
    p, f = buf.ctypes.data, ctypes.c_float
    m = _lib.matrix12(np.eye(3, 4))
    big, huge = 1 << 30, 1 << 62
    cases = [
        ("lsr_flatfield_apply_f32", (p, p, p, p, huge, 4, 4, None)),
        ("lsr_flatfield_pattern_f32", (p, 8, big, big, p, p, p, None)),
        ("lsr_deskew_f32", (p, 1 << 20, 1 << 20, 1 << 20, p, 4, 4, 4, 4, 16, 4, m, 1, None)),
        ("lsr_deskew_f32", (p, 8, 8, 8, p, 4, 4, 4, 1 << 40, 1 << 50, 4, m, 1, None)),
        ("lsr_affine_f32", (p, big, 4, 4, p, 4, 4, 4, m, f(0.0), 0, None)),
        ("lsr_affine_pitched_f32", (p, 8, 8, 8, 1 << 31, 1 << 40, p, 8, 8, 8, 8, 64, m, f(0.0), 0, None)),
        ("lsr_correlate_sep_f32", (p, p + 64, None, huge, huge, 4, p, 3, p, 3, p, 3, 0, f(1e-6), None, None, None, None)),
        ("lsr_minmax_f32", (p, 1 << 50, p, p, None)),
        ("lsr_match_shape_f32", (p, 4, 4, 4, p + 64, big, 4, 4, None)),
        ("lsr_deskew_f32_cpu", (p, 1 << 20, 1 << 20, 1 << 20, p, 4, 4, 4, 4, 16, 4, m, 1, None)),
        ("lsr_correlate_sep_f32_cpu", (p, p + 64, None, huge, 4, 4, p, 3, p, 3, p, 3, 0, f(1e-6), None, None, None, None)),
        ("lsr_blur_reflect_f32_cpu", (p, p + 64, big, 4, 4, 0, p, 1, f(0.0), f(0.0), None)),
    ]
    for name, args in cases:
        rc = getattr(lib, name)(*args)
        assert rc == _lib.E_UNSUPPORTED, (name, rc, lib.lsr_last_error())
    assert lib.lsr_affine_path(huge, huge, huge, m, 0) == 0
    lib.lsr_rfft_rows_scratch_bytes.restype = ctypes.c_int64
    assert lib.lsr_rfft_rows_scratch_bytes(ctypes.c_int64(huge), ctypes.c_int64(huge)) == -1


def test_fourier_domain_rl_grid_and_method_arguments():
    """``deconvolve_fft.fft_grid``: per axis max(n + p // 2, p) rounded up to a length the transform kernels take (z and
    y 5-smooth, x four times a 5-smooth number, at least 2 x . x 8); ``make_plan`` refuses an unknown method before it
    touches a device."""
    from shrimpy_amd.deconvolve import make_plan
    from shrimpy_amd.deconvolve_fft import _next_smooth, fft_grid

    assert [_next_smooth(n) for n in (1, 7, 11, 171, 178, 2055, 2057)] == [1, 8, 12, 180, 180, 2160, 2160]
    assert fft_grid((171, 2048, 2270), (9, 7, 7)) == (180, 2160, 2304)          # config 2, the declared PSF's extents
    assert fft_grid((171, 2048, 2270), (31, 37, 19)) == (192, 2160, 2304)       # a 30 x 36 x 18 bead patch, padded to odd
    assert fft_grid((1, 35, 86), (1, 9, 25)) == (2, 40, 100)                    # single plane: the z leg transforms >= 2 points
    assert fft_grid((2, 1, 3), (3, 1, 1)) == (3, 1, 8)                          # the x leg transforms >= 8 points
    assert fft_grid((5, 6, 9), (9, 13, 17)) == (9, 15, 20)                      # thinner than the PSF: at least the PSF itself
    for (shape, psf) in (((86, 2048, 2491), (15, 19, 19)), ((67, 2048, 2540), (15, 19, 19))):   # configs 4 and 5
        g = fft_grid(shape, psf)
        assert all(a >= n + p // 2 for a, n, p in zip(g, shape, psf)) and g[0] <= 256 and g[2] <= 4096 and g[2] % 4 == 0
    with pytest.raises(ValueError, match="method"):
        make_plan((8, 8, 8), np.ones((3, 3, 3), np.float32) / 27, "cpu", method="spectral")
