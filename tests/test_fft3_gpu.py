"""The axis-by-axis 3-D real FFT behind ``dynatrack._phase_cross_corr`` (``shrimpy_amd/fft3.py``):
``lsr_transpose_last2_c64`` against ``torch.permute``, the transforms against ``torch.fft.rfftn`` /
``irfftn`` (float32 FFT tolerance, stated below), and the tracker's shifts through both routes."""
import numpy as np
import pytest
import torch

from shrimpy_amd import _lib, fft3
from shrimpy_amd import dynatrack as d

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.mark.parametrize("shape", [(3, 64, 64), (2, 70, 33), (1, 180, 1000), (5, 1, 129), (1, 7, 1), (4, 200, 65)])
def test_transpose_last2_equals_permute(shape):
    a, b, c = shape
    src = torch.randn((a, b, c), dtype=torch.complex64, device=DEV)
    dst = torch.full((a, c, b), complex(7, 7), dtype=torch.complex64, device=DEV)
    _lib.call("lsr_transpose_last2_c64", src.data_ptr(), dst.data_ptr(), a, b, c, _lib.stream_ptr(torch.device(DEV)))
    torch.cuda.synchronize()
    assert torch.equal(dst, src.permute(0, 2, 1).contiguous())
    with pytest.raises(_lib.LsrError, match="out of place"):
        _lib.call("lsr_transpose_last2_c64", src.data_ptr(), src.data_ptr(), a, b, c, _lib.stream_ptr(torch.device(DEV)))


@pytest.mark.parametrize("shape", [(12, 32, 48), (9, 20, 15), (180, 64, 50), (6, 125, 36), (1, 8, 8)])
def test_rfft3_and_irfft3_against_torch_fft(shape):
    """Same library underneath (rocFFT): agreement to float32 FFT rounding, 2e-6 of the largest
    coefficient; the input is left untouched; the inverse is N times ``irfftn``."""
    if not fft3.available():
        pytest.skip("hipFFT's C API is not loadable from this PyTorch")
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.rand(shape, device=DEV, generator=g) * 900 + 80
    keep = x.clone()
    spec = fft3.rfft3(x)
    assert torch.equal(x, keep)
    ref = torch.fft.rfftn(x)
    assert tuple(spec.shape) == tuple(reversed(ref.shape)) and spec.is_contiguous()
    err = (spec.permute(2, 1, 0) - ref).abs().max() / ref.abs().max()
    assert float(err) < 2e-6
    n = float(np.prod(shape))
    back = fft3.irfft3(spec.clone(), shape)
    assert tuple(back.shape) == tuple(shape)
    assert float((back / n - x).abs().max()) < 2e-6 * float(x.abs().max()) * np.log2(n)
    want = torch.fft.irfftn(ref, s=shape)
    assert float((back / n - want).abs().max()) < 2e-6 * float(x.abs().max()) * np.log2(n)
    with pytest.raises(ValueError):
        fft3.rfft3(x[:, ::2])                      # not contiguous
    with pytest.raises(ValueError):
        fft3.irfft3(spec, (shape[0] + 1,) + tuple(shape[1:]))


def test_tracker_shifts_are_the_same_through_both_fft_routes():
    """``_phase_cross_corr`` on rolled volumes (the reference's own test: rolled by (1, 2, -3),
    ``test_dynatrack.py:102-111``) and on a non-5-smooth grid: the axis-by-axis route and
    ``torch.fft.rfftn`` / ``irfftn`` find the same peak; spectra of the two routes never mix in the cache."""
    if not fft3.available():
        pytest.skip("hipFFT's C API is not loadable from this PyTorch")
    rng = np.random.default_rng(42)
    cases = [((8, 32, 32), (1, 2, -3)), ((17, 45, 70), (-2, 5, 4)), ((30, 64, 96), (3, -7, 11))]
    try:
        for shape, roll in cases:
            ref = torch.as_tensor(rng.random(shape).astype(np.float32), device=DEV)
            mov = torch.roll(ref, shifts=roll, dims=(0, 1, 2))
            d.set_spectrum_cache_bytes(1 << 30)
            d._axis_fft_ok[0] = True
            a = d._phase_cross_corr(ref, mov)
            a2 = d._phase_cross_corr(ref, mov)             # second call: the cached [XC][Y][Z] spectrum
            d._axis_fft_ok[0] = False
            b = d._phase_cross_corr(ref, mov)              # torch.fft route; must not pick up the other layout
            assert a == a2 == b
            if shape == (8, 32, 32):
                assert a == (1, 2, -3)
    finally:
        d._axis_fft_ok[0] = True
        d.set_spectrum_cache_bytes(0)


@pytest.mark.parametrize("shape", [(12, 32, 48), (9, 20, 15), (180, 70, 50), (90, 33, 36), (72, 64, 130), (256, 40, 24),
                                   (2, 8, 8), (250, 31, 18)])
def test_z_leg_in_one_kernel_equals_the_five_pass_route(shape):
    """``correlate_with_spectrum`` (x and y transforms, then ``lsr_cross_correlate_z_c64``: forward z
    transform, product with the reference spectrum, inverse z transform in LDS) against
    ``irfft3(ref * conj(rfft3(mov)))`` and against ``torch.fft``: float32 FFT rounding, stated below.
    z lengths with every radix (4, 2, 3, 5), ragged y tiles, one- and many-tile grids."""
    if not fft3.available():
        pytest.skip("hipFFT's C API is not loadable from this PyTorch")
    assert _lib.call_value("lsr_cross_correlate_z_supported", shape[0]) == 1
    g = torch.Generator(device=DEV).manual_seed(7)
    ref = torch.rand(shape, device=DEV, generator=g) * 900 + 80
    mov = torch.roll(ref, shifts=(1, -2, 3), dims=(0, 1, 2)) + torch.rand(shape, device=DEV, generator=g)
    spec = fft3.rfft3(ref)
    keep = spec.clone()
    got = fft3.correlate_with_spectrum(spec, mov)
    assert torch.equal(spec, keep)                                          # the cached spectrum is left intact
    prod = spec * torch.conj(fft3.rfft3(mov))
    five = fft3.irfft3(prod.contiguous(), shape)
    n = float(np.prod(shape))
    want = torch.fft.irfftn(torch.fft.rfftn(ref) * torch.conj(torch.fft.rfftn(mov)), s=shape) * n
    scale = float(want.abs().max())
    assert float((got - five).abs().max()) < 4e-6 * scale
    assert float((got - want).abs().max()) < 4e-6 * scale
    assert int(torch.argmax(got)) == int(torch.argmax(want))


def test_z_lengths_the_one_kernel_route_declines():
    for n, ok in ((1, 0), (2, 1), (7, 0), (171, 0), (180, 1), (256, 1), (270, 0), (243, 1), (250, 1)):
        assert _lib.call_value("lsr_cross_correlate_z_supported", n) == ok, n


ROW_CASES = [((12, 32, 48), (12, 32, 48)), ((9, 20, 16), (10, 24, 20)), ((171, 60, 50), (180, 64, 60)),
             ((8, 33, 130), (8, 36, 128)), ((30, 17, 2270), (30, 18, 2304)), ((5, 9, 8), (6, 10, 8)),
             ((20, 70, 4000), (20, 72, 4096))]


@pytest.mark.parametrize("src_shape,grid", ROW_CASES)
def test_row_kernels_against_torch_fft(src_shape, grid):
    """``spectrum_of`` (reflect pad / crop + x transform + transposes in this package's kernels) against
    ``torch.fft.rfftn`` of the ``_match_shape``-d volume, and ``correlation_peak`` against the argmax of the
    five-pass correlation: float32 FFT rounding (2e-6 of the largest coefficient), identical peak."""
    if not fft3.rows_supported(grid):
        pytest.skip("hipFFT's C API is not loadable from this PyTorch")
    g = torch.Generator(device=DEV).manual_seed(3)
    ref = torch.rand(src_shape, device=DEV, generator=g) * 900 + 80
    matched = d._match_shape(ref, grid)
    spec = fft3.spectrum_of(ref, grid)
    want = torch.fft.rfftn(matched)
    assert tuple(spec.shape) == tuple(reversed(want.shape))
    assert float((spec.permute(2, 1, 0) - want).abs().max() / want.abs().max()) < 2e-6
    mov = torch.roll(ref, shifts=(1, -2, 3), dims=(0, 1, 2)) + torch.rand(src_shape, device=DEV, generator=g)
    idx = int(fft3.correlation_peak(spec, mov, grid).item())
    corr = torch.fft.irfftn(want * torch.conj(torch.fft.rfftn(d._match_shape(mov, grid))), s=grid)
    shifted = torch.fft.fftshift(corr.abs())
    assert idx == int(torch.argmax(shifted))
    # and through the tracker: same shift as the torch.fft route
    try:
        a = d._phase_cross_corr(ref, mov)
        d._axis_fft_ok[0] = False
        assert d._phase_cross_corr(ref, mov) == a
    finally:
        d._axis_fft_ok[0] = True


def test_row_kernels_decline_what_they_do_not_take():
    assert _lib.call_value("lsr_rfft_rows_supported", 2304) == 1 and _lib.call_value("lsr_rfft_rows_supported", 4096) == 1
    for n in (6, 90, 2270, 4100, 8192, 2):          # not a multiple of 4 / half not 5-smooth / too long
        assert _lib.call_value("lsr_rfft_rows_supported", n) == 0, n
    assert fft3.rows_supported((180, 2048, 2304)) == fft3.available()
    assert not fft3.rows_supported((171, 2048, 2304))            # z length 171 is not 5-smooth: five-pass z leg
