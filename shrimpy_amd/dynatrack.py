"""DynaTrack shift estimators on the deskewed volume, on MI355X.

Mirrors the private estimator functions of the reference's ``shrimpy/dynatrack/tracking.py`` --
same names, arguments, return types and corner-case behaviour.  The arithmetic runs as HIP kernels (``csrc/estimators.hip``: min / max, ``torch.histc``-exact
histograms, fp64 centroid sums, LDS-tiled reflect-padded Gaussian passes); the 256-bin searches
(percentile, multi-Otsu) are a few hundred flops of host logic on the histogram, like the
reference's own ``float(...)`` / ``int(...)`` round trips; the three FFTs of the phase
cross-correlation are rocFFT library calls through ``torch.fft``.

=========================================  =====================================  ==========
reference (``tracking.py``)                here                                   kernels
=========================================  =====================================  ==========
``_gaussian_blur_3d`` ``:386-422``         :func:`_gaussian_blur_3d`              blur x3
``_multiotsu_threshold`` ``:425-501``      :func:`_multiotsu_threshold`           minmax, histogram
``_binary_mask`` ``:504-542``              :func:`_binary_mask`                   minmax, blur x3, histogram
``_center_of_mass`` ``:545-569``           :func:`_center_of_mass`                mask centroid
``_percentile`` ``:572-593``               :func:`_percentile`                    minmax, histogram
``_intensity_center_of_mass`` ``:596-649`` :func:`_intensity_center_of_mass`      weighted centroid
``..._to_roi_center`` ``:652-707``         same name                              the above
``_multiotsu_center_of_mass`` ``:759-787`` same name                              the above
``_match_shape`` ``:266-306``              :func:`_match_shape`                   match shape
``_phase_cross_corr`` ``:309-378``         :func:`_phase_cross_corr`              match shape, cross power, peak (+ rocFFT)
``_roi_center_pcc`` / ``_multiotsu_pcc``   same names                             the above
=========================================  =====================================  ==========

``_binary_mask`` / ``_center_of_mass`` never materialise the boolean mask unless asked to: the
centroid kernel thresholds on the fly.

CPU tensors -- the reference's tracker runs on ``cuda`` if available, else ``cpu`` (``tracking.py:1054``), and its CI
has no GPU -- take the same functions through the native host twins of those kernels (``csrc/estimators_host.hip``,
``lsr_*_cpu``: same signatures, same arithmetic) and ``torch.fft`` on the CPU for the transforms; nothing here reads
``oracle/``.  A volume is processed where it lives; nothing is moved between devices behind the caller's back.
"""

from __future__ import annotations

import ctypes
import logging

import numpy as np

from . import _lib

logger = logging.getLogger(__name__)

__all__ = [
    "_gaussian_blur_3d", "_multiotsu_threshold", "_binary_mask", "_center_of_mass", "_percentile",
    "_intensity_center_of_mass", "_intensity_center_of_mass_to_roi_center", "_multiotsu_center_of_mass",
    "_next_fast_len", "_match_shape", "_phase_cross_corr", "_centered_gaussian_blob", "_roi_center_pcc",
    "_multiotsu_pcc", "set_spectrum_cache_bytes", "invalidate_reference", "ShiftSettings", "SegmentationSettings", "RoiCenterSettings",
    "_limit_shifts_zyx", "compute_shift", "TRACKING_METHODS",
]


def _volume(img, name="img"):
    import torch

    if not isinstance(img, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor, got {type(img).__name__}")
    if img.dtype != torch.float32:
        img = img.to(dtype=torch.float32)
    if img.device.type == "cpu":
        return img.contiguous()
    return _lib.require_device_f32(img.contiguous(), name)


def _scratch(device):
    import torch

    return torch.empty((_lib.call_value("lsr_reduce_scratch_bytes"),), dtype=torch.uint8, device=device)


def _run(device, entry: str, *args) -> None:
    """Call ``entry`` for operands on ``device``: the kernel on its current stream, or -- CPU tensors -- the
    entry's host twin (``entry + "_cpu"``, same arguments; worker threads = ``torch.get_num_threads()``)."""
    import torch

    if device.type == "cpu":
        from . import host

        host._threads()
        _lib.call(entry + "_cpu", *args, None)
        return
    with torch.cuda.device(device):
        _lib.call(entry, *args, _lib.stream_ptr(device))


def _minmax(img) -> tuple[float, float]:
    """``(float(img.min()), float(img.max()))`` -- one pass, one host round trip (the reference's
    ``float(vmin)`` / ``float(vmax)`` synchronise too)."""
    import torch

    out = torch.empty((2,), dtype=torch.float32, device=img.device)
    _run(img.device, "lsr_minmax_f32", img.data_ptr(), img.numel(), out.data_ptr(), _scratch(img.device).data_ptr())
    lo, hi = out.cpu().tolist()
    return lo, hi


_HIST_PIECE = 1 << 31     # voxels per lsr_histogram_f32 call (< 2**32: the bins count in 32 bits)


def _histc(img, nbins: int, vmin: float, vmax: float) -> np.ndarray:
    """``torch.histc(img, bins=nbins, min=vmin, max=vmax)`` as a host float32 array."""
    import torch

    # the bins are uint32 counters: a volume of 2**32 voxels or more is histogrammed in pieces (counts add up)
    flat = img.reshape(-1)
    total = np.zeros((nbins,), dtype=np.int64)
    counts = torch.empty((nbins,), dtype=torch.int32, device=img.device)
    for a in range(0, flat.numel(), _HIST_PIECE):
        piece = flat[a:a + _HIST_PIECE]
        _run(img.device, "lsr_histogram_f32", piece.data_ptr(), piece.numel(), ctypes.c_float(vmin), ctypes.c_float(vmax),
             int(nbins), counts.data_ptr())
        total += counts.cpu().numpy().view(np.uint32)
    return total.astype(np.float32)


def _gaussian_blur_3d(img, sigma: float, _rescale: tuple[float, float] | None = None):
    """Separable 3-D Gaussian blur with reflect padding (``tracking.py:386-422``).

    ``_rescale = (vmin, vmax)`` fuses the reference's ``(img - vmin) / (vmax - vmin)`` into the
    first pass.
    """
    import torch

    vol = _volume(img)
    if vol.dim() != 3:
        raise ValueError(f"img must be (Z, Y, X), got shape {tuple(vol.shape)}")
    if sigma <= 0:
        if _rescale is not None:
            return (vol - _rescale[0]) / (_rescale[1] - _rescale[0])
        return vol
    max_radius = int(4 * sigma + 0.5)
    z, y, x = (int(v) for v in vol.shape)
    src = vol
    sub, div = (0.0, 0.0) if _rescale is None else (float(_rescale[0]), float(np.float32(_rescale[1]) - np.float32(_rescale[0])))
    for axis, n in enumerate((z, y, x)):
        r = min(max_radius, n - 1)   # reflect padding requires pad < dim (reference :403-404)
        xs = torch.arange(-r, r + 1, device=vol.device, dtype=torch.float32)
        k1d = torch.exp(-0.5 * (xs / sigma) ** 2)
        k1d = (k1d / k1d.sum()).contiguous()
        dst = torch.empty_like(vol)
        _run(vol.device, "lsr_blur_reflect_f32", src.data_ptr(), dst.data_ptr(), z, y, x, axis, k1d.data_ptr(), r,
             ctypes.c_float(sub), ctypes.c_float(div))
        src, sub, div = dst, 0.0, 0.0
    return src


def _otsu_from_hist(hist: np.ndarray, vmin: float, vmax: float, otsu_component: int) -> float:
    """The 3-class search of ``tracking.py:466-501`` on a host histogram (float32 throughout)."""
    nbins = hist.shape[0]
    hist = (hist / hist.sum(dtype=np.float32)).astype(np.float32)
    # torch.linspace(start, end, steps) in float32: start + i * step for the first half,
    # end - (steps - 1 - i) * step for the second (ATen's symmetric formula)
    step = (np.float32(vmax) - np.float32(vmin)) / np.float32(nbins - 1)
    idx = np.arange(nbins, dtype=np.float32)
    half = nbins // 2
    centers = np.where(np.arange(nbins) < half, np.float32(vmin) + step * idx,
                       np.float32(vmax) - step * (np.float32(nbins - 1) - idx)).astype(np.float32)
    cum_w = np.cumsum(hist, dtype=np.float32)
    cum_wm = np.cumsum(hist * centers, dtype=np.float32)
    total_mean = cum_wm[-1]
    eps = np.float32(1e-10)
    w0 = cum_w[:, None]
    w1 = cum_w[None, :] - cum_w[:, None]
    w2 = np.float32(1.0) - cum_w[None, :]
    m0 = cum_wm[:, None] / np.maximum(w0, eps)
    m1 = (cum_wm[None, :] - cum_wm[:, None]) / np.maximum(w1, eps)
    m2 = (total_mean - cum_wm[None, :]) / np.maximum(w2, eps)
    sigma = w0 * (m0 - total_mean) ** 2 + w1 * (m1 - total_mean) ** 2 + w2 * (m2 - total_mean) ** 2
    bins = np.arange(nbins)
    valid = ((bins[None, :] > bins[:, None]) & (bins[None, :] <= nbins - 2)
             & (w0 > eps) & (w1 > eps) & (w2 > eps))
    sigma = np.where(valid, sigma, np.float32(-1.0)).astype(np.float32)
    best_a, best_b = divmod(int(np.argmax(sigma)), nbins)
    thresholds = (float(centers[best_a + 1]), float(centers[best_b + 1]))
    logger.debug("multi-Otsu thresholds: %s (using component %d)", thresholds, otsu_component)
    return thresholds[min(otsu_component, 1)]


def _multiotsu_threshold(img_blur, otsu_component: int = 0, nbins: int = 256) -> float:
    """Multi-Otsu threshold of a (pre-blurred) device volume (``tracking.py:425-501``)."""
    vol = _volume(img_blur, "img_blur")
    vmin, vmax = _minmax(vol)
    if vmin == vmax:
        return float(vmin)
    return _otsu_from_hist(_histc(vol, nbins, vmin, vmax), vmin, vmax, otsu_component)


def _binary_mask(img, sigma: float = 5.0, otsu_component: int = 0):
    """Rescale to [0, 1], blur, multi-Otsu threshold (``tracking.py:504-542``) -> boolean mask."""
    import torch

    vol = _volume(img)
    blurred, threshold = _blurred_and_threshold(vol, sigma, otsu_component)
    if blurred is None:
        return torch.zeros_like(vol, dtype=torch.bool)
    return blurred > threshold


def _blurred_and_threshold(vol, sigma: float, otsu_component: int):
    vmin, vmax = _minmax(vol)
    if not vmax > vmin:
        return None, 0.0
    blurred = _gaussian_blur_3d(vol, sigma, _rescale=(vmin, vmax))
    return blurred, _multiotsu_threshold(blurred, otsu_component)


def _centroid(kernel: str, vol, param: float):
    import torch

    z, y, x = (int(v) for v in vol.shape)
    out = torch.empty((4,), dtype=torch.float64, device=vol.device)
    _run(vol.device, kernel, vol.data_ptr(), z, y, x, ctypes.c_float(param), out.data_ptr(),
         _scratch(vol.device).data_ptr())
    return out.cpu().numpy()


def _center_of_mass(mask, _threshold: float | None = None):
    """Centre of mass of a boolean mask (``tracking.py:545-569``): every True voxel counts once.

    Given a float volume and ``_threshold`` it is the centroid of ``volume > _threshold`` without
    the mask ever being written.
    """
    import torch

    if _threshold is None:
        if mask.dtype != torch.bool:
            raise TypeError("mask must be a boolean tensor")
        vol, thr = _volume(mask.to(torch.float32), "mask"), 0.5
    else:
        vol, thr = _volume(mask, "mask"), float(_threshold)
    if vol.dim() != 3:
        raise ValueError(f"mask must be (Z, Y, X), got shape {tuple(vol.shape)}")
    s = _centroid("lsr_mask_centroid_f32", vol, thr)
    if s[0] == 0:
        return torch.zeros(3, device=vol.device)
    return torch.as_tensor((s[1:] / s[0]).astype(np.float32), device=vol.device)


def _percentile(img, percentile: float, nbins: int = 256) -> float:
    """Percentile (0-100) from a 256-bin histogram: upper edge of the selected bin (``:572-593``)."""
    vol = _volume(img)
    vmin, vmax = _minmax(vol)
    if vmax <= vmin:
        return vmin
    hist = _histc(vol, nbins, vmin, vmax)
    cdf = np.cumsum(hist, dtype=np.float32)
    cdf = cdf / cdf[-1]
    idx = int(np.searchsorted(cdf, np.float32(percentile / 100.0), side="left"))
    idx = min(idx, nbins - 1)
    return vmin + (idx + 1) * (vmax - vmin) / nbins


def _intensity_center_of_mass(img, background: float = 0.0):
    """Intensity-weighted centre of mass, weights ``max(img - background, 0)`` (``:596-649``)."""
    import torch

    vol = _volume(img)
    if vol.dim() != 3:
        raise ValueError(f"img must be (Z, Y, X), got shape {tuple(vol.shape)}")
    s = _centroid("lsr_weighted_centroid_f32", vol, float(background))
    if s[0] <= 0:
        # no positive mass: the geometric centre, so that a ROI-centre shift is zero (:631-640)
        return torch.tensor([(n - 1) / 2.0 for n in vol.shape], device=vol.device, dtype=torch.float32)
    return torch.as_tensor((s[1:] / s[0]).astype(np.float32), device=vol.device)


def _intensity_center_of_mass_to_roi_center(current_img, background_percentile: float | None = None,
                                            blur_sigma: float = 0.0) -> tuple[float, ...]:
    """Shift from the ROI centre to the intensity-weighted centroid, ZYX pixels (``:652-707``)."""
    img = _volume(current_img, "current_img")
    if blur_sigma and blur_sigma > 0:
        img = _gaussian_blur_3d(img, blur_sigma)
    background = _percentile(img, background_percentile) if background_percentile is not None else 0.0
    com = _intensity_center_of_mass(img, background=background).cpu().numpy()
    roi_center = np.array([(s - 1) / 2.0 for s in img.shape], dtype=np.float32)
    shift = com - roi_center
    logger.debug("intensity_center_of_mass: com=%s roi_center=%s background=%.4g blur_sigma=%.2g shift=%s",
                 com.tolist(), roi_center.tolist(), background, blur_sigma, shift.tolist())
    return tuple(float(s) for s in shift)


def _multiotsu_center_of_mass(ref_img, mov_img, sigma: float = 5.0, otsu_component: int = 0) -> tuple[float, ...]:
    """Shift between the multi-Otsu mask centroids of two volumes, ZYX pixels (``:759-787``)."""
    centers = []
    for k, img in enumerate((ref_img, mov_img)):
        vol = _volume(img)
        tag = ("multiotsu_center", float(sigma), int(otsu_component))
        cacheable = k == 0 and vol is img and _spectra.max_bytes > 0   # the stored reference tensor
        center = _spectra.get(vol, tag) if cacheable else None
        if center is None:
            blurred, threshold = _blurred_and_threshold(vol, sigma, otsu_component)
            if blurred is None:
                center = np.zeros(3, np.float32)        # empty mask -> origin (:559-560)
            else:
                center = _center_of_mass(blurred, _threshold=threshold).cpu().numpy()
            if cacheable:
                _spectra.put(vol, tag, center)
        centers.append(center)
    shift = centers[1] - centers[0]
    logger.debug("multiotsu_center_of_mass: ref_center=%s mov_center=%s shift=%s", centers[0].tolist(),
                 centers[1].tolist(), shift.tolist())
    return tuple(float(s) for s in shift)


class _ReferenceCache:
    """Results derived from a reference volume, kept between timepoints.

    The updater compares every timepoint of a position with one stored reference tensor
    (``tracking.py:1119-1151``), so whatever an estimator computes from the reference alone -- its
    spectrum (a third of ``_phase_cross_corr``), its multi-Otsu mask or that mask's centroid (half
    of ``_multiotsu_center_of_mass``) -- repeats identical work.  Entries are keyed by the tensor
    OBJECT (weak reference; the entries go when the tensor does) plus a tag, and checked against
    the tensor's version counter, storage and shape, so an in-place edit or a different tensor is a
    miss, never a stale hit.  This package's own kernels write through raw pointers, which the version
    counter does not see: every entry point that takes ``out=`` therefore bumps it
    (``_lib.mark_written``); a caller that rewrites a reference buffer by other means calls
    :func:`invalidate_reference`.  Byte-bounded LRU over the device tensors it holds (4 GiB by
    default: the spectrum of one config-2 reference); ``set_spectrum_cache_bytes(0)`` turns it off --
    the reference itself keeps nothing and calls ``empty_cache()`` after every correlation
    (``tracking.py:1097-1102``).
    """

    def __init__(self, max_bytes: int):
        from collections import OrderedDict

        self.max_bytes = int(max_bytes)
        self._entries: "OrderedDict[tuple, tuple]" = OrderedDict()
        self._bytes = 0
        self.hits = self.misses = 0

    @staticmethod
    def _nbytes(value) -> int:
        return value.numel() * value.element_size() if hasattr(value, "element_size") else 0

    def _drop(self, key):
        e = self._entries.pop(key, None)
        if e is not None:
            self._bytes -= self._nbytes(e[-1])

    def _drop_tensor(self, tid):
        for key in [k for k in self._entries if k[0] == tid]:
            self._drop(key)

    def clear(self):
        self._entries.clear()
        self._bytes = 0

    def get(self, t, tag):
        key = (id(t), tag)
        e = self._entries.get(key)
        if e is not None:
            ref, version, ptr, shape, value = e
            if ref() is t and version == t._version and ptr == t.data_ptr() and shape == tuple(t.shape):
                self._entries.move_to_end(key)
                self.hits += 1
                return value
            self._drop_tensor(id(t))
        self.misses += 1
        return None

    def put(self, t, tag, value):
        import weakref

        nbytes = self._nbytes(value)
        if self.max_bytes <= 0 or nbytes > self.max_bytes:
            return
        key = (id(t), tag)
        self._drop(key)
        while self._entries and self._bytes + nbytes > self.max_bytes:
            self._drop(next(iter(self._entries)))
        self._entries[key] = (weakref.ref(t, lambda _r, tid=id(t): self._drop_tensor(tid)), t._version,
                              t.data_ptr(), tuple(t.shape), value)
        self._bytes += nbytes


_spectra = _ReferenceCache(4 << 30)


def set_spectrum_cache_bytes(n: int) -> None:
    """Upper bound on device memory held by cached reference results (default 4 GiB; 0 = off)."""
    _spectra.max_bytes = int(n)
    _spectra.clear()


def invalidate_reference(t=None) -> None:
    """Forget what was cached for the reference tensor ``t`` (all references when ``None``): call it
    after refreshing a reference volume in place by a route torch's version counter cannot see."""
    if t is None:
        _spectra.clear()
    else:
        _spectra._drop_tensor(id(t))


def _next_fast_len(n: int) -> int:
    """Smallest 5-smooth integer >= n (``tracking.py:248-263``)."""
    if n <= 1:
        return 1
    while True:
        m = n
        for p in (2, 3, 5):
            while m % p == 0:
                m //= p
        if m == 1:
            return n
        n += 1


def _match_shape(t, shape):
    """Reflect-pad or centre-crop ``t`` to ``shape``, per axis (``tracking.py:266-306``)."""
    import torch

    vol = _volume(t, "t")
    shape = tuple(int(v) for v in shape)
    if vol.dim() != len(shape) or vol.dim() not in (2, 3):
        raise ValueError(f"cannot match a {tuple(vol.shape)} tensor to shape {shape}: two or three axes each")
    if vol.dim() == 2:      # images (``test_dynatrack.py:62-82``): a one-plane volume
        return _match_shape(vol.unsqueeze(0), (1,) + shape).squeeze(0)
    if tuple(vol.shape) == shape:
        return vol
    out = torch.empty(shape, dtype=torch.float32, device=vol.device)
    _run(vol.device, "lsr_match_shape_f32", vol.data_ptr(), *(int(v) for v in vol.shape), out.data_ptr(), *shape)
    return out


def _spectrum(vol, shape, kind: str):
    """Forward transform of ``vol`` matched to the FFT ``shape``: ``kind`` "rfft3" = one batched
    1-D library transform per axis with this package's transposes in between (``fft3.rfft3``,
    spectrum laid out ``[XC][Y][Z]``), "rfftn" = ``torch.fft.rfftn``."""
    import torch

    matched = _match_shape(vol, shape)
    if kind == "rfft3":
        from . import fft3

        return fft3.rfft3(matched)
    return torch.fft.rfftn(matched)


_axis_fft_ok = [True]      # cleared when hipFFT refuses a plan: the torch.fft route is used from then on
_rows_ok = [True]       # the LDS row / z transforms launched (cleared on the first refusal)


def _phase_cross_corr(ref_img, mov_img, maximum_shift: float = 1.0) -> tuple[int, ...]:
    """FFT phase cross-correlation, pixel shifts in ZYX order (``tracking.py:309-378``).

    The transforms are rocFFT library calls; the steps around them -- shape matching,
    ``f1 * conj(f2)``, ``argmax(fftshift(|corr|))`` -- are HIP kernels that write nothing but their
    result.  The 3-D transforms run axis by axis (``fft3``: contiguous batched 1-D transforms through
    hipFFT's C API, this package's transposes in between, spectra kept in the transposed layout);
    ``torch.fft.rfftn`` / ``irfftn`` are the fallback when hipFFT's C API is not loadable.
    """
    import torch

    from . import fft3

    if (isinstance(ref_img, torch.Tensor) and isinstance(mov_img, torch.Tensor)
            and ref_img.dim() == 2 and mov_img.dim() == 2):
        # 2-D images (the reference takes both, ``test_dynatrack.py:85-100``): a one-plane volume; the
        # length-1 axis transforms to itself and contributes shift 0
        return _phase_cross_corr(ref_img.unsqueeze(0), mov_img.unsqueeze(0), maximum_shift)[1:]
    ref_t, mov_t = _volume(ref_img, "ref_img"), _volume(mov_img, "mov_img")
    if ref_t.dim() != 3 or mov_t.dim() != 3:
        raise ValueError("phase cross-correlation takes two (Y, X) images or two (Z, Y, X) volumes")
    shape = tuple(_next_fast_len(int(max(s1, s2) * maximum_shift)) for s1, s2 in zip(ref_t.shape, mov_t.shape))
    logger.debug("phase cross corr: fft shape %s for arrays %s and %s (max_shift=%.2f)", shape,
                 tuple(ref_t.shape), tuple(mov_t.shape), maximum_shift)
    if ref_t.device != mov_t.device:
        raise ValueError(f"ref_img is on {ref_t.device}, mov_img on {mov_t.device}")
    on_gpu = ref_t.device.type == "cuda"      # (CPU tensors: torch.fft between the host twins of the other steps)
    kind = "rfft3" if (on_gpu and _axis_fft_ok[0] and fft3.available() and min(shape) >= 2) else "rfftn"
    peak_index = None
    try:
        if kind == "rfft3" and _rows_ok[0] and fft3.rows_supported(shape):
            try:
                peak_index = _correlation_peak_rows(ref_t, ref_t is ref_img, mov_t, shape)
            except _lib.LsrError as exc:
                # the LDS transforms were refused at launch (their 78-139 KB of dynamic LDS only fit a
                # gfx950 CU): keep tracking through the library transforms instead of failing the volume
                logger.warning("LDS row / z transforms unavailable (%s): using the hipFFT route", exc)
                _rows_ok[0] = False
                corr = _cross_correlation(ref_t, ref_t is ref_img, mov_t, shape, kind)
        else:
            corr = _cross_correlation(ref_t, ref_t is ref_img, mov_t, shape, kind)
    except fft3.AxisFftError as exc:
        logger.warning("axis-by-axis FFT unavailable (%s): using torch.fft", exc)
        _axis_fft_ok[0] = False
        peak_index = None
        corr = _cross_correlation(ref_t, ref_t is ref_img, mov_t, shape, "rfftn")
    if peak_index is None:
        peak_index = torch.empty((1,), dtype=torch.int64, device=corr.device)
        _run(corr.device, "lsr_peak_abs_shifted_f32", corr.data_ptr(), *shape, peak_index.data_ptr(),
             _scratch(corr.device).data_ptr())
    peak = np.unravel_index(int(peak_index.item()), shape)
    result = tuple(int(s // 2) - int(p) for s, p in zip(shape, peak))
    logger.debug("phase cross corr: peak at %s (device=%s)", result, ref_t.device)
    return result


def _correlation_peak_rows(ref_t, ref_is_callers, mov_t, shape):
    """The whole correlation with this package's kernels along x and z (``fft3.correlation_peak``): the flat
    index of the peak; the reference's spectrum is cached under the same key as the other axis-by-axis route."""
    from . import fft3

    cacheable = ref_is_callers and _spectra.max_bytes > 0
    fimg1 = _spectra.get(ref_t, ("rfft3", shape)) if cacheable else None
    if fimg1 is None:
        fimg1 = fft3.spectrum_of(ref_t, shape)
        if cacheable:
            _spectra.put(ref_t, ("rfft3", shape), fimg1)
    return fft3.correlation_peak(fimg1, mov_t, shape)


def _same_dense_layout(a, b) -> bool:
    """Same shape and strides, and those strides a permutation of a contiguous tensor's (no gaps, no overlap)."""
    if a.shape != b.shape or a.stride() != b.stride() or a.dtype != b.dtype:
        return False
    expect = 1
    for size, stride in sorted(zip(a.shape, a.stride()), key=lambda p: p[1]):
        if size == 1:
            continue
        if stride != expect:
            return False
        expect *= size
    return True


def _cross_correlation(ref_t, ref_is_callers, mov_t, shape, kind: str):
    """``irfftn(rfftn(ref) * conj(rfftn(mov)))`` on the FFT grid ``shape`` (any positive scale: only
    its peak is used), float32 ``(Z, Y, X)``."""
    import torch

    # the reference's spectrum is reused while the caller keeps comparing against the same tensor
    cacheable = ref_is_callers and _spectra.max_bytes > 0
    fimg1 = _spectra.get(ref_t, (kind, shape)) if cacheable else None
    if fimg1 is None:
        fimg1 = _spectrum(ref_t, shape, kind)
        if cacheable:
            _spectra.put(ref_t, (kind, shape), fimg1)
    if kind == "rfft3" and _lib.call_value("lsr_cross_correlate_z_supported", shape[0]):
        # short z axis (a deskewed volume's always is): forward z transform, product and inverse z
        # transform in one kernel on LDS-resident columns (csrc/zcorr.hip)
        from . import fft3

        return fft3.correlate_with_spectrum(fimg1, _match_shape(mov_t, shape))
    fimg2 = _spectrum(mov_t, shape, kind)
    # f1 * conj(f2), written over f2: f1 may be the cached spectrum.  Element-wise over the raw storage, so the two
    # must share one dense layout -- they do (the same transform of the same shape: rocFFT's rfftn hands back a
    # permuted, dense spectrum, and both come back that way); anything else is brought to the standard layout first
    if not _same_dense_layout(fimg1, fimg2):
        fimg1, fimg2 = fimg1.contiguous(), fimg2.contiguous()
    _run(fimg2.device, "lsr_cross_power_into_c64", fimg1.data_ptr(), fimg2.data_ptr(), fimg2.numel())
    del fimg1
    if kind == "rfft3":
        from . import fft3

        return fft3.irfft3(fimg2, shape)
    return torch.fft.irfftn(fimg2, s=shape).contiguous()


def _centered_gaussian_blob(shape, sigma: float, device):
    """Separable Gaussian blob centred on the geometric centre (``tracking.py:710-732``)."""
    import torch

    axes_1d = []
    for n in shape:
        idx = torch.arange(n, device=device, dtype=torch.float32)
        axes_1d.append(torch.exp(-0.5 * ((idx - (n - 1) / 2.0) / sigma) ** 2))
    blob = axes_1d[0]
    for g in axes_1d[1:]:
        blob = blob.unsqueeze(-1) * g
    return blob


_blobs: dict = {}


def _roi_center_pcc(current_img, blob_sigma: float = 10.0, maximum_shift: float = 1.0) -> tuple[int, ...]:
    """Shift of the bright structure from the ROI centre: PCC against a centred blob (``:735-756``)."""
    img = _volume(current_img, "current_img")
    # one blob object per (shape, sigma, device): its spectrum is then reused like a reference's
    key = (tuple(img.shape), float(blob_sigma), str(img.device))
    blob = _blobs.get(key)
    if blob is None:
        if len(_blobs) >= 2:
            _blobs.pop(next(iter(_blobs)))
        blob = _blobs[key] = _centered_gaussian_blob(tuple(img.shape), blob_sigma, img.device)
    return _phase_cross_corr(blob, img, maximum_shift)


def _multiotsu_pcc(ref_img, mov_img, sigma: float = 5.0, otsu_component: int = 0,
                   maximum_shift: float = 1.0) -> tuple[int, ...]:
    """PCC on the multi-Otsu masks of two volumes (``tracking.py:790-815``)."""
    import torch

    tag = ("multiotsu_mask", float(sigma), int(otsu_component))
    cacheable = isinstance(ref_img, torch.Tensor) and _spectra.max_bytes > 0
    ref_mask = _spectra.get(ref_img, tag) if cacheable else None
    if ref_mask is None:
        ref_mask = _binary_mask(ref_img, sigma=sigma, otsu_component=otsu_component).to(dtype=torch.float32)
        if cacheable:   # the same mask object next time: its spectrum is then cached as well
            _spectra.put(ref_img, tag, ref_mask)
    mov_mask = _binary_mask(mov_img, sigma=sigma, otsu_component=otsu_component).to(dtype=torch.float32)
    return _phase_cross_corr(ref_mask, mov_mask, maximum_shift)


# ---------------------------------------------------------------------------------------------
# The dispatcher on top of the estimators (DynaTrackUpdater._compute_shift, tracking.py:1224-1312)
# ---------------------------------------------------------------------------------------------

from pydantic import BaseModel, ConfigDict  # noqa: E402


class ShiftSettings(BaseModel):
    """How the raw shift estimate is searched, bounded and scaled (``tracking.py:45-66``)."""

    model_config = ConfigDict(extra="forbid")

    maximum: float = 1.0
    limits: dict[str, tuple[float, float]] | None = None
    dampening: tuple[float, float, float] | None = None


class SegmentationSettings(BaseModel):
    """Parameters of the ``multiotsu_*`` methods (``tracking.py:69-83``)."""

    model_config = ConfigDict(extra="forbid")

    otsu_sigma: float = 5.0
    otsu_component: int = 0


class RoiCenterSettings(BaseModel):
    """Parameters of the referenceless ROI-centre methods (``tracking.py:86-112``)."""

    model_config = ConfigDict(extra="forbid")

    blob_sigma: float = 10.0
    background_percentile: float | None = None
    blur_sigma: float = 0.0


TRACKING_METHODS = ("pcc", "intensity_center_of_mass", "roi_center_pcc", "multiotsu_center_of_mass",
                    "multiotsu_pcc")


def _limit_shifts_zyx(shifts_zyx, shift_limits: dict) -> np.ndarray:
    """Per-axis deadband and clip in microns (``tracking.py:822-868``): below the minimum -> 0, above
    the maximum -> the maximum with the shift's sign."""
    shifts_zyx = np.array(shifts_zyx, dtype=float)
    for i, axis in enumerate(("z", "y", "x")):
        if axis not in shift_limits:
            continue
        min_limit, max_limit = shift_limits[axis]
        if abs(shifts_zyx[i]) < min_limit:
            shifts_zyx[i] = 0.0
        elif abs(shifts_zyx[i]) > max_limit:
            shifts_zyx[i] = np.sign(shifts_zyx[i]) * max_limit
    return shifts_zyx


def compute_shift(reference_zyx, current_zyx, tracking_method: str = "pcc", *,
                  shift: ShiftSettings | dict | None = None,
                  segmentation: SegmentationSettings | dict | None = None,
                  roi_center: RoiCenterSettings | dict | None = None,
                  scale_z: float = 1.0, scale_yx: float = 1.0) -> tuple[float, float, float]:
    """Shift of ``current_zyx`` against ``reference_zyx`` as (x, y, z) in microns: the body of
    ``DynaTrackUpdater._compute_shift`` (``tracking.py:1224-1312``) -- method dispatch, pixels to
    microns, limits, dampening, (z, y, x) -> (x, y, z) -- on device tensors."""
    def model(cls, v):
        return v if isinstance(v, cls) else cls(**(v or {}))

    sh, seg, roi = model(ShiftSettings, shift), model(SegmentationSettings, segmentation), \
        model(RoiCenterSettings, roi_center)
    method = tracking_method
    if method == "pcc":
        px = _phase_cross_corr(reference_zyx, current_zyx, sh.maximum)
    elif method == "intensity_center_of_mass":
        px = _intensity_center_of_mass_to_roi_center(current_zyx, background_percentile=roi.background_percentile,
                                                     blur_sigma=roi.blur_sigma)
    elif method == "roi_center_pcc":
        px = _roi_center_pcc(current_zyx, blob_sigma=roi.blob_sigma, maximum_shift=sh.maximum)
    elif method == "multiotsu_center_of_mass":
        px = _multiotsu_center_of_mass(reference_zyx, current_zyx, sigma=seg.otsu_sigma,
                                       otsu_component=seg.otsu_component)
    elif method == "multiotsu_pcc":
        px = _multiotsu_pcc(reference_zyx, current_zyx, sigma=seg.otsu_sigma, otsu_component=seg.otsu_component,
                            maximum_shift=sh.maximum)
    else:
        raise ValueError(f"Unknown tracking_method={method!r}. Use 'pcc', 'intensity_center_of_mass', "
                         "'roi_center_pcc', 'multiotsu_center_of_mass', or 'multiotsu_pcc'.")
    um = np.array([px[0] * scale_z, px[1] * scale_yx, px[2] * scale_yx], dtype=float)
    if sh.limits is not None:
        um = _limit_shifts_zyx(um, sh.limits)
    if sh.dampening is not None:
        um = um * np.array(sh.dampening, dtype=float)
    return float(um[2]), float(um[1]), float(um[0])
