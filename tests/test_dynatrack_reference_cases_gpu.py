"""The estimator cases the reference's own test-suite holds (``shrimpy/tests/test_dynatrack.py:38-111,
589-812``: shapes, known answers and inequalities on constructed volumes), run through this package's
estimators on the device.  The inputs are rebuilt from the descriptions there (seeded generators,
boxes set to constants); the assertions are the reference's expected outcomes."""
import numpy as np
import pytest
import torch

from shrimpy_amd import dynatrack as d

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=DEV)


def _box(shape, fill=0.0, boxes=()):
    v = np.full(shape, fill, dtype=np.float32)
    for (z0, z1, y0, y1, x0, x1), val, add in boxes:
        if add:
            v[z0:z1, y0:y1, x0:x1] += val
        else:
            v[z0:z1, y0:y1, x0:x1] = val
    return v


def test_match_shape_cases_2d():
    """``TestMatchShape`` (:62-82): pad smaller, crop larger, pad one axis while cropping the other -- 2-D inputs."""
    for shape in ((2, 3), (6, 7), (2, 7)):
        out = d._match_shape(torch.ones(shape, device=DEV), (4, 5))
        assert tuple(out.shape) == (4, 5) and bool((out == 1).all())
    t = torch.arange(20, dtype=torch.float32, device=DEV).reshape(4, 5)
    assert torch.equal(d._match_shape(t, (4, 5)), t)
    np.testing.assert_array_equal(d._match_shape(t, (2, 3)).cpu().numpy(), t.cpu().numpy()[1:3, 1:4])   # centre crop


def test_gaussian_blur_cases():
    """``TestGaussianBlur3D`` (:589-606)."""
    g = torch.Generator(device=DEV).manual_seed(0)
    vol = torch.rand((8, 32, 32), device=DEV, generator=g)
    assert d._gaussian_blur_3d(vol, sigma=2.0).shape == vol.shape
    out = d._gaussian_blur_3d(vol, sigma=3.0)
    assert float(out.max() - out.min()) <= float(vol.max() - vol.min())
    assert torch.equal(d._gaussian_blur_3d(vol, sigma=0.0), vol)


def test_binary_mask_cases():
    """``TestBinaryMask`` (:609-629): boolean mask with some voxels set; a higher Otsu component is stricter."""
    rng = np.random.default_rng(42)
    vol = rng.random((8, 32, 32)) * 0.2
    vol[3:6, 12:20, 12:20] = 0.9
    mask = d._binary_mask(_dev(vol), sigma=1.0, otsu_component=0)
    assert mask.dtype == torch.bool and mask.device.type == torch.device(DEV).type and int(mask.sum()) > 0
    rng = np.random.default_rng(42)
    vol = rng.random((8, 32, 32)) * 0.3
    vol[2:6, 8:24, 8:24] = 0.6
    vol[3:5, 12:20, 12:20] = 0.95
    m0 = d._binary_mask(_dev(vol), sigma=1.0, otsu_component=0)
    m1 = d._binary_mask(_dev(vol), sigma=1.0, otsu_component=1)
    assert int(m0.sum()) >= int(m1.sum())


def test_center_of_mass_cases():
    """``TestCenterOfMass`` (:632-645): a centred cube -> (4.5, 4.5, 4.5); an empty mask -> zeros."""
    mask = torch.zeros((10, 10, 10), dtype=torch.bool, device=DEV)
    mask[3:7, 3:7, 3:7] = True
    assert torch.allclose(d._center_of_mass(mask).cpu(), torch.tensor([4.5, 4.5, 4.5]), atol=0.5)
    empty = torch.zeros((10, 10, 10), dtype=torch.bool, device=DEV)
    assert torch.equal(d._center_of_mass(empty).cpu().float(), torch.zeros(3))


def test_intensity_center_of_mass_cases():
    """``TestIntensityCenterOfMass`` (:648-693)."""
    cube = ((3, 7, 3, 7, 3, 7), 1.0, False)
    c = d._intensity_center_of_mass(_dev(_box((10, 10, 10), 0.0, [cube]))).cpu()
    assert torch.allclose(c.float(), torch.tensor([4.5, 4.5, 4.5]), atol=0.1)
    bright = _box((10, 10, 10), 0.0, [cube])
    bright[3:7, 3:7, 6] = 100.0
    assert float(d._intensity_center_of_mass(_dev(bright))[2]) > 4.5
    blank = d._intensity_center_of_mass(torch.zeros((10, 10, 10), device=DEV)).cpu()
    assert torch.allclose(blank.float(), torch.tensor([4.5, 4.5, 4.5]))         # geometric centre, not the origin
    neg = _box((10, 10, 10), 0.0, [cube])
    neg[0, 0, 0] = -1000.0
    assert torch.allclose(d._intensity_center_of_mass(_dev(neg)).cpu().float(), torch.tensor([4.5, 4.5, 4.5]), atol=0.1)
    ped = _dev(_box((8, 64, 64), 0.2, [((2, 6, 44, 52, 28, 36), 5.0, True)]))
    assert float(d._intensity_center_of_mass(ped, background=0.2)[1]) > float(d._intensity_center_of_mass(ped)[1])


def test_percentile_cases():
    """``TestPercentile`` (:696-707): the median of a 0..999 ramp within a bin; a flat image returns its value."""
    ramp = torch.arange(1000, dtype=torch.float32, device=DEV).reshape(10, 10, 10)
    assert abs(d._percentile(ramp, 50.0) - 499.5) < 1000 / 256 + 1
    assert d._percentile(torch.full((4, 8, 8), 3.0, device=DEV), 90.0) == pytest.approx(3.0)


def test_roi_centre_shift_cases():
    """``TestIntensityCenterOfMassToRoiCenter`` (:710-743)."""
    centred = _dev(_box((8, 64, 64), 0.0, [((2, 6, 30, 34, 30, 34), 1.0, False)]))
    assert all(abs(s) < 1.0 for s in d._intensity_center_of_mass_to_roi_center(centred))
    off = _dev(_box((8, 64, 64), 0.0, [((2, 6, 40, 50, 28, 36), 1.0, False)]))
    assert d._intensity_center_of_mass_to_roi_center(off)[1] > 0
    assert all(abs(s) < 1e-3 for s in d._intensity_center_of_mass_to_roi_center(torch.zeros((8, 64, 64), device=DEV)))
    flat = d._intensity_center_of_mass_to_roi_center(torch.full((8, 64, 64), 3.0, device=DEV), background_percentile=99.0)
    assert all(abs(s) < 1e-3 for s in flat)


def test_centred_blob_and_roi_centre_pcc_cases():
    """``TestCenteredGaussianBlob`` / ``TestRoiCenterPcc`` (:746-771)."""
    blob = d._centered_gaussian_blob((8, 32, 32), sigma=4.0, device=torch.device(DEV))
    peak = np.unravel_index(int(torch.argmax(blob)), tuple(blob.shape))
    assert peak[0] in (3, 4) and peak[1] in (15, 16) and peak[2] in (15, 16)
    near = d._roi_center_pcc(_dev(_box((8, 64, 64), 0.0, [((2, 6, 28, 36, 28, 36), 1.0, False)])), blob_sigma=4.0)
    assert abs(near[1]) <= 2 and abs(near[2]) <= 2
    off = d._roi_center_pcc(_dev(_box((8, 64, 64), 0.0, [((2, 6, 44, 52, 28, 36), 1.0, False)])), blob_sigma=4.0)
    assert off[1] > 4


def test_multiotsu_tracker_cases():
    """``TestMultiotsuCenterOfMass`` / ``TestMultiotsuPcc`` (:774-811): one generator feeds both volumes in turn."""
    rng = np.random.default_rng(42)
    ref = rng.random((8, 64, 64)) * 0.1
    ref[2:6, 20:40, 20:40] = 0.9
    mov = rng.random((8, 64, 64)) * 0.1
    mov[2:6, 25:45, 20:40] = 0.9
    s = d._multiotsu_center_of_mass(_dev(ref), _dev(mov), sigma=1.0, otsu_component=0)
    assert abs(s[1] - 5.0) < 2.0 and abs(s[2]) < 2.0 and abs(s[0]) < 2.0
    rng = np.random.default_rng(42)
    ref = rng.random((8, 64, 64)) * 0.1
    ref[2:6, 15:50, 15:50] = 0.9
    mov = rng.random((8, 64, 64)) * 0.1
    mov[2:6, 18:53, 15:50] = 0.9
    s = d._multiotsu_pcc(_dev(ref), _dev(mov), sigma=1.0, otsu_component=0)
    assert abs(s[1] - 3) <= 1 and abs(s[2]) <= 1


def test_compute_shift_cases():
    """``TestComputeShift`` (:219-298): pixels -> microns, dampening, limits, on rng(42) (8, 64, 64) rolled volumes."""
    ref = _dev(np.random.default_rng(42).random((8, 64, 64)))
    mov = torch.roll(ref, shifts=(1, 2, -3), dims=(0, 1, 2))
    x, y, z = d.compute_shift(ref, mov, "pcc", scale_yx=0.5, scale_z=2.0)
    assert (x, y, z) == pytest.approx((-3 * 0.5, 2 * 0.5, 1 * 2.0), abs=1e-6)
    x, y, z = d.compute_shift(ref, mov, "pcc", scale_yx=0.5, scale_z=2.0, shift={"dampening": (0.5, 0.25, 0.1)})
    assert (x, y, z) == pytest.approx((-3 * 0.5 * 0.1, 2 * 0.5 * 0.25, 1 * 2.0 * 0.5), abs=1e-6)
    mov = torch.roll(ref, shifts=-3, dims=2)
    limits = {"z": (0.1, 50.0), "y": (0.1, 50.0), "x": (0.1, 5.0)}
    x, y, z = d.compute_shift(ref, mov, "pcc", scale_yx=10.0, scale_z=2.0, shift={"limits": limits})
    assert x == pytest.approx(-5.0, abs=1e-6) and z == pytest.approx(0.0, abs=1e-6)     # -30 um clipped, sign kept
    with pytest.raises(ValueError, match="Unknown tracking_method"):
        d.compute_shift(ref, mov, "nope")
