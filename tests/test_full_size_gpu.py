"""Parity at BASELINE config 2's full size: raw (2048, 512, 2048) -> (171, 2048, 2270) -> 20-iteration RL.

The oracle cannot run the whole volume in test time, so it runs where the domain lets a part stand
for the whole: the deskew is independent per raw-X column (Y' is raw X reversed; the reference
chunks along it, ``scripts/measure_psf.py:221-249``), and after n RL iterations a voxel depends
only on the input within n * 2 * (PSF radius) of it -- so the oracle deskews column slabs and
deconvolves crops with that margin, and the device result of the full-size run must match there.
The rest are properties that hold at any size: the one-launch and two-launch RL agree bit for bit,
flux weighted by the border normalisation is conserved, the uint16 input path equals the float one.
"""

import numpy as np
import pytest

from oracle import cpu_ref as o

pytestmark = pytest.mark.gpu

RAW_SHAPE = (2048, 512, 2048)
DESKEW = dict(ls_angle_deg=30.0, px_to_scan_ratio=0.755, keep_overhang=False, average_n_slices=3)
PSF_SHAPE, PSF_SIGMA = (9, 7, 7), (2.0, 1.2, 1.2)


@pytest.fixture(scope="module")
def scene(device):
    """Raw bead scene (seed per SURVEY 8(d): 1000 * config + 7 * position) and its deskew, on the device."""
    import torch

    import bench
    from shrimpy_amd.deskew import fast_deskew_zyx

    raw = bench.synthetic_raw(RAW_SHAPE, seed=2000, device=device)
    deskewed = fast_deskew_zyx(raw_data=raw, **DESKEW)
    assert tuple(deskewed.shape) == (171, 2048, 2270)
    yield raw, deskewed
    del raw, deskewed
    torch.cuda.empty_cache()


def test_deskew_full_size_matches_the_oracle_on_raw_x_slabs(scene):
    raw, deskewed = scene
    X = RAW_SHAPE[2]
    for a, b in ((0, 3), (1021, 1025), (X - 2, X)):
        slab = raw[:, :, a:b].contiguous().cpu().numpy()
        want = o.deskew(slab, DESKEW["ls_angle_deg"], DESKEW["px_to_scan_ratio"], DESKEW["keep_overhang"],
                        DESKEW["average_n_slices"])
        got = deskewed[:, X - b:X - a, :].cpu().numpy()
        np.testing.assert_array_equal(got, want)


def test_deskew_full_size_uint16_input_gives_the_same_bits(scene, device):
    import torch

    from shrimpy_amd.deskew import fast_deskew_zyx

    raw, deskewed = scene
    assert float(raw.max()) < 65536 and bool((raw == raw.round()).all())  # Poisson counts
    raw16 = raw.to(torch.uint16)
    assert torch.equal(fast_deskew_zyx(raw_data=raw16, **DESKEW), deskewed)


@pytest.fixture(scope="module")
def deconvolved(scene, device):
    from shrimpy_amd.deconvolve import RichardsonLucyPlan

    _, deskewed = scene
    factors = o.gaussian_psf(PSF_SHAPE, PSF_SIGMA)[1]
    plan = RichardsonLucyPlan(tuple(deskewed.shape), None, device, psf_factors=factors)
    assert plan.fused
    out = plan(deskewed, iterations=20)
    return plan, factors, out


def test_rl_full_size_one_launch_equals_two_launch_bit_for_bit(scene, deconvolved, device):
    import torch

    from shrimpy_amd.deconvolve import RichardsonLucyPlan

    _, deskewed = scene
    _, factors, fused = deconvolved
    plan = RichardsonLucyPlan(tuple(deskewed.shape), None, device, psf_factors=factors, fused="never")
    two = plan(deskewed, iterations=20)
    plan.release()
    assert torch.equal(fused, two)


def test_rl_full_size_is_finite_non_negative_and_conserves_weighted_flux(scene, deconvolved):
    """sum_v x_new(v) * (H^T 1)(v) = sum_u y(u) * (Hx)(u) / ((Hx)(u) + eps) ~= sum y."""
    import torch

    _, deskewed = scene
    plan, _, x = deconvolved
    assert bool(torch.isfinite(x).all()) and float(x.min()) >= 0
    nz, ny, nx = (n.double() for n in plan._norm)
    flux = torch.einsum("zyx,z,y,x->", x.double(), nz, ny, nx)
    total = deskewed.double().sum()
    assert abs(float(flux / total) - 1.0) < 1e-4


def test_rl_full_size_scalars_from_the_epilogue(scene, deconvolved, device):
    """VERDICT r3 row g at config 2: the fused launch's own sums -- flux of every iteration within 1e-4 of sum y, the
    last iteration's total / change equal to torch's fp64 reductions of the estimates (rel 1e-5), the two-launch form
    the same scalars, the estimate untouched, and tol stops the loop where the relative change says."""
    import torch

    from shrimpy_amd.deconvolve import RichardsonLucyPlan

    _, deskewed = scene
    plan, factors, x20 = deconvolved
    x = plan(deskewed, iterations=20, stats=True)
    assert torch.equal(x, x20)
    s = plan.last_stats
    total_y = float(deskewed.double().sum())
    assert np.all(np.abs(s.flux / total_y - 1.0) < 1e-4), s.flux / total_y
    x19 = plan(deskewed, iterations=19)
    assert abs(s.total[19] / float(x20.double().sum()) - 1.0) < 1e-5
    assert abs(s.total[18] / float(x19.double().sum()) - 1.0) < 1e-5
    assert abs(s.change[19] / float((x20.double() - x19.double()).abs().sum()) - 1.0) < 1e-5
    assert np.all(np.diff(s.rel_change) < 0), "the relative change of RL on a bead scene falls monotonically"
    del x19
    two = RichardsonLucyPlan(tuple(deskewed.shape), None, device, psf_factors=factors, fused="never")
    two(deskewed, iterations=3, stats=True)
    for name in ("flux", "change", "total"):
        np.testing.assert_allclose(getattr(two.last_stats, name), getattr(s, name)[:3], rtol=1e-6, err_msg=name)
    two.release()
    tol = float(0.5 * (s.rel_change[9] + s.rel_change[10]))
    xt = plan(deskewed, iterations=20, tol=tol)
    st = plan.last_stats
    assert st.stopped_by_tol and st.iterations in (11, 12), st.iterations
    assert torch.equal(xt, plan(deskewed, iterations=st.iterations))


@pytest.mark.parametrize("where", ["interior", "corner"])
def test_rl_full_size_matches_the_oracle_on_a_crop_with_full_margin(scene, deconvolved, where):
    """20 iterations reach 20 * 2 * (4, 3, 3) = (160, 120, 120) voxels: all of z, 120 in the plane."""
    _, deskewed = scene
    _, factors, x = deconvolved
    margin, core = 120, 48
    if where == "interior":
        y0, x0 = 900, 1300
        crop = (slice(None), slice(y0 - margin, y0 + core + margin), slice(x0 - margin, x0 + core + margin))
        inner = (slice(None), slice(margin, margin + core), slice(margin, margin + core))
        full = (slice(None), slice(y0, y0 + core), slice(x0, x0 + core))
    else:  # the volume's own (y, x) = (0, 0) corner: real borders on two sides, margin on the others
        crop = (slice(None), slice(0, core + margin), slice(0, core + margin))
        inner = (slice(None), slice(0, core), slice(0, core))
        full = inner
    y_crop = deskewed[crop].contiguous().cpu().numpy()
    want = o.richardson_lucy_separable(y_crop, factors, iterations=20)[inner].astype(np.float64)
    got = x[full].cpu().numpy().astype(np.float64)
    tol = 2e-4 * np.abs(want) + 1e-4 * np.abs(want).max()   # the RL bar of tests/test_gpu_parity.py
    assert np.all(np.abs(got - want) <= tol), float(np.max(np.abs(got - want) / tol))


# ---------------------------------------------------------------- config 3: registration apply


def _config3_matrix():
    """rotation 2 deg about Z, scale (1, .98, 1.02), translation (3.5,-12.25,20.75)."""
    th = np.deg2rad(2.0)
    rot = np.array([[1, 0, 0], [0, np.cos(th), -np.sin(th)], [0, np.sin(th), np.cos(th)]])
    m = np.eye(4)
    m[:3, :3] = rot @ np.diag([1.0, 0.98, 1.02])
    m[:3, 3] = [3.5, -12.25, 20.75]
    return m


def test_affine_full_size_matches_the_oracle_on_output_blocks(device):
    """(256, 2048, 2048) moving volume, config-3 matrix: scipy evaluates output blocks from the
    source box they reach (cropped where that box is inside the volume, cut at the volume's own
    faces where it is not).  The block-local coordinates differ from the whole-grid ones in the last
    fp64 bit, so a float32 result may round the other way: one ulp is allowed."""
    import torch

    from shrimpy_amd.register import apply_affine_transform_zyx

    shape = (256, 2048, 2048)
    g = torch.Generator(device=device).manual_seed(3000)
    moving = torch.rand(shape, device=device, generator=g) * 1000 - 100
    m = _config3_matrix()
    out = apply_affine_transform_zyx(moving, m, shape, cval=-3.0)
    blocks = [((100, 1000, 900), (8, 40, 60)),      # interior
              ((0, 0, 0), (8, 40, 60)),             # source coordinates below 0 in y: cval region
              ((250, 2000, 1980), (6, 48, 68))]     # far corner: beyond the last source voxels
    for origin, size in blocks:
        origin, size = np.array(origin), np.array(size)
        corners = np.array([[origin[i] + (size[i] - 1) * ((c >> i) & 1) for i in range(3)] for c in range(8)])
        src = corners @ m[:3, :3].T + m[:3, 3]
        lo = np.maximum(np.floor(src.min(0)).astype(int) - 2, 0)
        hi = np.minimum(np.ceil(src.max(0)).astype(int) + 3, np.array(shape))
        if np.any(hi - lo < 2):   # the block sees nothing of the volume
            assert bool((out[tuple(slice(a, a + n) for a, n in zip(origin, size))] == -3.0).all())
            continue
        crop = moving[tuple(slice(a, b) for a, b in zip(lo, hi))].contiguous().cpu().numpy()
        offset = m[:3, :3] @ origin + m[:3, 3] - lo
        # cut faces of the crop that are not faces of the volume must be out of the block's reach
        assert np.all((lo == 0) | (src.min(0) - lo >= 1)) and np.all((hi == shape) | (hi - 1 - src.max(0) >= 1))
        want = o.affine_apply(crop, m[:3, :3], offset, tuple(size), cval=-3.0)
        got = out[tuple(slice(a, a + n) for a, n in zip(origin, size))].cpu().numpy()
        np.testing.assert_array_max_ulp(got, want, maxulp=1)
        assert np.mean(got == want) > 0.95
    del moving, out
    torch.cuda.empty_cache()


@pytest.mark.parametrize("tilted", [False, True])
def test_affine_full_size_grid_constant_matches_the_oracle_on_output_blocks(device, tilted):
    """The blending border rule at config-3 size through the LDS-staged kernels (planar for the config-3 map,
    box for the map with a 3 degree tilt about y on top): wherever the plain rule keeps a sample the two rules
    give the same bits; blocks at the volume's faces and inside it against scipy's ``mode="grid-constant"`` on
    the source box they reach (one ulp: block-local fp64 coordinates)."""
    import torch

    from shrimpy_amd import _lib
    from shrimpy_amd.geometry import as_matrix_3x4
    from shrimpy_amd.register import apply_affine_transform_zyx

    shape = (256, 2048, 2048)
    m = _config3_matrix()
    if tilted:
        c3, s3 = np.cos(np.deg2rad(3.0)), np.sin(np.deg2rad(3.0))
        m[:3, :3] = np.array([[c3, 0, -s3], [0, 1, 0], [s3, 0, c3]]) @ m[:3, :3]
    assert _lib.call_value("lsr_affine_path", *shape, _lib.matrix12(as_matrix_3x4(m)),
                           _lib.MODE_GRID_CONSTANT) == (2 if tilted else 1)
    g = torch.Generator(device=device).manual_seed(3002)
    moving = torch.rand(shape, device=device, generator=g) * 1000 - 100
    out = apply_affine_transform_zyx(moving, m, shape, cval=-3.0, mode="grid-constant")
    plain = apply_affine_transform_zyx(moving, m, shape, cval=-3.0)
    kept = plain != -3.0
    assert torch.equal(out[kept], plain[kept])
    assert int((out != plain).sum()) > 0          # ... and the blended shell exists
    del plain, kept
    blocks = [((100, 1000, 900), (8, 40, 60)), ((0, 0, 0), (8, 40, 60)), ((250, 2000, 1980), (6, 48, 68)),
              ((0, 1000, 0), (6, 40, 64)), ((120, 8, 2040), (9, 30, 8))]
    checked = 0
    for origin, size in blocks:
        origin, size = np.array(origin), np.array(size)
        corners = np.array([[origin[i] + (size[i] - 1) * ((c >> i) & 1) for i in range(3)] for c in range(8)])
        src = corners @ m[:3, :3].T + m[:3, 3]
        lo = np.maximum(np.floor(src.min(0)).astype(int) - 3, 0)
        hi = np.minimum(np.ceil(src.max(0)).astype(int) + 4, np.array(shape))
        sl = tuple(slice(a, a + n) for a, n in zip(origin, size))
        if np.any(hi - lo < 2):
            continue
        # cut faces of the crop that are not faces of the volume must be out of the block's reach (two voxels:
        # the blending rule looks one voxel past a face)
        if not (np.all((lo == 0) | (src.min(0) - lo >= 2)) and np.all((hi == shape) | (hi - 1 - src.max(0) >= 2))):
            continue
        crop = moving[tuple(slice(a, b) for a, b in zip(lo, hi))].contiguous().cpu().numpy()
        want = o.affine_apply(crop, m[:3, :3], m[:3, :3] @ origin + m[:3, 3] - lo, tuple(size), cval=-3.0,
                              mode="grid-constant")
        got = out[sl].cpu().numpy()
        np.testing.assert_array_max_ulp(got, want, maxulp=1)
        assert np.mean(got == want) > 0.95
        checked += 1
    assert checked >= 3
    del moving, out
    torch.cuda.empty_cache()


def test_affine_full_size_tilted_map_matches_the_oracle_on_output_blocks(device):
    """Config 3 with a 3 degree tilt about y on top of its registration -- a map that couples z with
    the plane, i.e. ``affine_box.hip`` at full size: output blocks against scipy run on the source box
    they reach (at most one ulp: block-local fp64 coordinates), exact and f32 modes, and the f32 mode
    keeps the exact mode's in / out-of-range decisions over the WHOLE volume."""
    import torch

    from shrimpy_amd import _lib
    from shrimpy_amd.geometry import as_matrix_3x4
    from shrimpy_amd.register import apply_affine_transform_zyx

    shape = (256, 2048, 2048)
    m = _config3_matrix()
    c3, s3 = np.cos(np.deg2rad(3.0)), np.sin(np.deg2rad(3.0))
    m[:3, :3] = np.array([[c3, 0, -s3], [0, 1, 0], [s3, 0, c3]]) @ m[:3, :3]
    assert _lib.call_value("lsr_affine_path", *shape, _lib.matrix12(as_matrix_3x4(m)), _lib.MODE_CONSTANT) == 2
    g = torch.Generator(device=device).manual_seed(3001)
    moving = torch.rand(shape, device=device, generator=g) * 1000 - 100
    out = apply_affine_transform_zyx(moving, m, shape, cval=-3.0)
    out32 = apply_affine_transform_zyx(moving, m, shape, cval=-3.0, exact=False)
    assert torch.equal(out == -3.0, out32 == -3.0)
    assert float((out - out32).abs().max()) <= 2e-5 * 1100
    blocks = [((100, 1000, 900), (8, 40, 60)), ((60, 700, 1200), (9, 33, 70)), ((200, 1500, 400), (8, 40, 60)),
              ((0, 0, 0), (8, 40, 60)), ((248, 2000, 1980), (8, 48, 68)), ((120, 8, 2040), (9, 30, 8))]
    checked = 0
    for origin, size in blocks:
        origin, size = np.array(origin), np.array(size)
        corners = np.array([[origin[i] + (size[i] - 1) * ((c >> i) & 1) for i in range(3)] for c in range(8)])
        src = corners @ m[:3, :3].T + m[:3, 3]
        lo = np.maximum(np.floor(src.min(0)).astype(int) - 2, 0)
        hi = np.minimum(np.ceil(src.max(0)).astype(int) + 3, np.array(shape))
        sl = tuple(slice(a, a + n) for a, n in zip(origin, size))
        if np.any(hi - lo < 2):
            assert bool((out[sl] == -3.0).all())
            continue
        if not (np.all((lo == 0) | (src.min(0) - lo >= 1)) and np.all((hi == shape) | (hi - 1 - src.max(0) >= 1))):
            continue
        crop = moving[tuple(slice(a, b) for a, b in zip(lo, hi))].contiguous().cpu().numpy()
        want = o.affine_apply(crop, m[:3, :3], m[:3, :3] @ origin + m[:3, 3] - lo, tuple(size), cval=-3.0)
        got = out[sl].cpu().numpy()
        np.testing.assert_array_max_ulp(got, want, maxulp=1)
        assert np.mean(got == want) > 0.95
        checked += 1
    assert checked >= 4
    del moving, out, out32
    torch.cuda.empty_cache()


# ---------------------------------------------------------------- the rows either side, full size


def test_flatfield_full_size_median_is_exact_on_row_slabs(scene):
    """Per-pixel median over 2048 raw planes: numpy sorts the columns of a few (y) rows."""
    from shrimpy_amd.flatfield import flat_field_pattern

    raw, _ = scene
    ff = flat_field_pattern(raw)
    pattern = ff.pattern.cpu().numpy()
    for y in (0, 255, 511):
        col = raw[:, y, :].cpu().numpy()
        srt = np.sort(col, axis=0)
        a, b = srt[col.shape[0] // 2 - 1], srt[col.shape[0] // 2]    # even Z: torch.quantile lerps
        np.testing.assert_array_equal(pattern[y], b - (b - a) * np.float32(0.5))
    assert float(ff.mean) == pytest.approx(float(pattern.astype(np.float64).mean()), rel=1e-6)


def test_estimators_full_size_match_the_oracle(scene):
    """Histogram percentile and intensity centroid of the whole (171, 2048, 2270) deskewed volume."""
    from shrimpy_amd import dynatrack as d

    _, deskewed = scene
    host = deskewed.cpu().numpy()
    for p in (50.0, 99.0):
        assert d._percentile(deskewed, p) == pytest.approx(o.dt_percentile(host, p), rel=1e-6)
    bg = o.dt_percentile(host, 90.0)
    np.testing.assert_allclose(d._intensity_center_of_mass(deskewed, bg).cpu().numpy(),
                               o.dt_intensity_center_of_mass(host, bg), atol=2e-3)


def test_rl_full_size_y_separable_path_agrees_with_the_dense_kernel(scene, device):
    """The secondary PSF (rotated 30 deg about Y) at config-2 size: (z, x) stencil + y pass against the
    441-tap dense kernel over the whole volume, and against the oracle on a crop with full margin."""
    import torch

    from shrimpy_amd.deconvolve import RichardsonLucyPlan

    _, deskewed = scene
    psf = o.rotated_psf(PSF_SHAPE, PSF_SIGMA, 30.0)
    iters = 6
    split = RichardsonLucyPlan(tuple(deskewed.shape), psf, device)
    assert split.path == "y-separable (fused)"
    a = split(deskewed, iterations=iters)
    split.release()
    two = RichardsonLucyPlan(tuple(deskewed.shape), psf, device, fused="never")
    assert two.path == "y-separable"
    assert torch.equal(a, two(deskewed, iterations=iters))      # the whole config-2 volume, bit for bit
    two.release()
    dense = RichardsonLucyPlan(tuple(deskewed.shape), psf, device, separable="never")
    assert dense.path == "dense"
    b = dense(deskewed, iterations=iters)
    dense.release()
    tol = 2e-4 * b.abs() + 1e-4 * float(b.max())
    assert bool(((a - b).abs() <= tol).all())
    margin, core = iters * 2 * 3, 40          # 6 iterations reach 36 voxels in the plane, all of z
    y0, x0 = 700, 1100
    crop = deskewed[:, y0 - margin:y0 + core + margin, x0 - margin:x0 + core + margin].contiguous().cpu().numpy()
    want = o.richardson_lucy(crop, psf, iters)[:, margin:margin + core, margin:margin + core].astype(np.float64)
    got = a[:, y0:y0 + core, x0:x0 + core].cpu().numpy().astype(np.float64)
    assert np.all(np.abs(got - want) <= 2e-4 * np.abs(want) + 1e-4 * np.abs(want).max())
    del a, b
    torch.cuda.empty_cache()


def test_rl_full_size_in_the_fourier_domain_agrees_with_the_dense_stencil_and_the_oracle(scene, device):
    """A dense 11 x 9 x 9 PSF at config-2 size through the Fourier-domain iteration (shrimpy_amd/deconvolve_fft.py,
    grid 180 x 2160 x 2304) against the tuned dense stencil over the WHOLE volume, against the oracle on a crop with
    full margin, and flux against sum y; then a bead-patch-sized 15 x 19 x 19 PSF, which only this route takes,
    against the oracle on a crop."""
    import torch

    import bench
    from shrimpy_amd.deconvolve import RichardsonLucyPlan, make_plan

    _, deskewed = scene
    psf = bench.measured_psf((11, 9, 9))
    iters = 4
    fft = make_plan(tuple(deskewed.shape), psf, device)
    assert fft.path == "fft" and fft.grid == (180, 2160, 2304)      # (891 taps: past the stencil's break-even)
    a = fft(deskewed, iterations=iters, stats=True)
    flux = fft.last_stats.flux
    fft.release()
    del fft
    dense = RichardsonLucyPlan(tuple(deskewed.shape), psf, device, separable="never")
    assert dense.path == "dense"
    b = dense(deskewed, iterations=iters)
    dense.release()
    del dense
    tol = 2e-4 * b.abs() + 1e-4 * float(b.max())
    assert bool(((a - b).abs() <= tol).all())
    np.testing.assert_allclose(flux, float(deskewed.sum(dtype=torch.float64)), rtol=1e-4)
    del b
    torch.cuda.empty_cache()

    def crop_check(got, kernel, n, y0, x0, core=32):
        ry, rx = kernel.shape[1] // 2, kernel.shape[2] // 2
        my, mx = n * 2 * ry, n * 2 * rx          # n iterations reach 2 n r voxels in the plane, all of z
        crop = deskewed[:, y0 - my:y0 + core + my, x0 - mx:x0 + core + mx].contiguous().cpu().numpy()
        want = o.richardson_lucy(crop, kernel, n, use_fft=True)[:, my:my + core, mx:mx + core].astype(np.float64)
        have = got[:, y0:y0 + core, x0:x0 + core].cpu().numpy().astype(np.float64)
        assert np.all(np.abs(have - want) <= 2e-4 * np.abs(want) + 1e-4 * np.abs(want).max())

    crop_check(a, psf, iters, 900, 1300)
    del a
    torch.cuda.empty_cache()
    patch = bench.measured_psf((15, 19, 19))
    plan = make_plan(tuple(deskewed.shape), patch, device)
    assert plan.path == "fft"
    c = plan(deskewed, iterations=2)
    plan.release()
    crop_check(c, patch, 2, 400, 700)
    crop_check(c, patch, 2, 0 + 2 * 2 * 9, 2270 - 32 - 2 * 2 * 9, core=32)      # near the y = 0 / x = X corner region
    del c
    torch.cuda.empty_cache()


@pytest.mark.parametrize("shape,grid", [((86, 2048, 2491), (96, 2160, 2500)), ((67, 2048, 2540), (75, 2160, 2560)),
                                        ((33, 1001, 1777), (40, 1024, 1800))])
def test_rl_in_the_fourier_domain_at_the_other_configs_sizes(device, shape, grid):
    """The deskewed sizes of BASELINE configs 4 and 5 (and an odd one): x legs of 1250 = 2 5^4, 1280 = 2^8 5 and
    900 = 2^2 3^2 5^2 points -- other radix mixes than config 2's 1152 = 2^7 3^2 -- two iterations with a bead-patch PSF
    against the oracle on crops with full margin, at a corner, in the middle and at the far corner."""
    import torch

    import bench
    from shrimpy_amd.deconvolve import make_plan

    psf = bench.measured_psf((15, 19, 19))
    g = torch.Generator(device=device).manual_seed(5)
    y = torch.poisson(torch.rand(shape, device=device, generator=g) * 300 + 50, generator=g)
    plan = make_plan(shape, psf, device)
    assert plan.path == "fft" and plan.grid == grid
    x = plan(y, iterations=2)
    plan.release()
    n, core = 2, 24
    my = mx = n * 2 * 9
    for (y0, x0) in ((my, mx), (shape[1] // 2, shape[2] // 2), (shape[1] - core - my, shape[2] - core - mx)):
        crop = y[:, y0 - my:y0 + core + my, x0 - mx:x0 + core + mx].contiguous().cpu().numpy()
        want = o.richardson_lucy(crop, psf, n, use_fft=True)[:, my:my + core, mx:mx + core].astype(np.float64)
        got = x[:, y0:y0 + core, x0:x0 + core].cpu().numpy().astype(np.float64)
        assert np.all(np.abs(got - want) <= 2e-4 * np.abs(want) + 1e-4 * np.abs(want).max())
    del x, y, plan
    torch.cuda.empty_cache()
