"""Richardson-Lucy 3-D deconvolution on MI355X.

No reference symbol exists (``docs/data_structure.md:58-62``: "algorithms for deconvolution ...
are being developed"); the north-star defines the step: 20 iterations of

    x <- x * H^T( y / (H x + eps) ) / (H^T 1)

with a 3-D PSF stencil and zero-padded borders, CPU path = explicit loop over
``scipy.ndimage.convolve`` / ``correlate`` (``oracle/cpu_ref.py:richardson_lucy``, test-only).

Each iteration is two launches of the z-marching LDS stencil ``csrc/correlate.hip`` with fused
epilogues (ratio; multiplicative update with analytic ``H^T 1``): 12 algorithmic bytes per voxel
per launch.  Rank-1 (separable) PSFs run ``pz+py+px`` FMAs per voxel and are HBM-bound; dense
PSFs run ``pz*py*px`` FMAs per voxel and are fp32-VALU-bound.  float32 throughout; agreement
with the float64-accumulating scipy loop is stated in ``tests/test_gpu_parity.py``.
CPU tensors (no HIP device in play) run the native host twins of the same launches (``shrimpy_amd/host.py``).
"""

from __future__ import annotations

import ctypes

from dataclasses import dataclass

import numpy as np

from . import _lib

__all__ = ["richardson_lucy", "RichardsonLucyPlan", "RLStats", "factor_psf", "correlate3d", "prepare_psf",
           "padded_shape", "PaddedVolume", "make_plan"]

MAX_TAPS = 15
MAX_Z_TAPS = 31     # separable PSFs only: the z factor runs as its own launch (csrc/correlate_z.hip)


def prepare_psf(psf, max_in_plane: int = MAX_TAPS, max_z: int = MAX_Z_TAPS) -> np.ndarray:
    """float32 (pz, py, px) with every axis odd: even axes get one trailing zero plane.

    With ``center = size // 2`` the padded kernel produces the same correlation.
    """
    p = np.asarray(psf, dtype=np.float32)
    if p.ndim != 3:
        raise ValueError(f"psf must be 3-D (Z, Y, X), got shape {p.shape}")
    if not np.all(np.isfinite(p)):
        raise ValueError("psf contains non-finite values")
    pad = [(0, 1 - (s % 2)) for s in p.shape]
    if any(hi for _, hi in pad):
        p = np.pad(p, pad)
    if max(p.shape[1:]) > max_in_plane or p.shape[0] > max_z:
        raise ValueError(f"psf shape {p.shape} exceeds {max_in_plane} taps in plane / {max_z} along z")
    return np.ascontiguousarray(p)


def factor_psf(psf, rtol: float = 1e-6):
    """Rank-1 factorisation ``psf ~= kz x ky x kx`` or ``None``.

    Accepted when ``max|psf - kz*ky*kx| <= rtol * max|psf|`` (float64 check).  Factors are
    returned as float32 with the overall scale carried by ``kz``.
    """
    p = np.asarray(psf, dtype=np.float64)
    pz, py, px = p.shape
    peak = np.abs(p).max()
    if peak == 0:
        return None
    u, s, vt = np.linalg.svd(p.reshape(pz, py * px), full_matrices=False)
    kz = u[:, 0] * s[0]
    yx = vt[0].reshape(py, px)
    u2, s2, vt2 = np.linalg.svd(yx, full_matrices=False)
    ky = u2[:, 0] * s2[0]
    kx = vt2[0]
    # fix signs so the in-plane factors are predominantly positive
    if ky.sum() < 0:
        ky, kz = -ky, -kz
    if kx.sum() < 0:
        kx, kz = -kx, -kz
    # balance: in-plane factors sum to 1, scale stays in kz
    sy, sx = ky.sum(), kx.sum()
    if sy != 0 and sx != 0:
        ky, kx, kz = ky / sy, kx / sx, kz * sy * sx
    approx = kz[:, None, None] * ky[None, :, None] * kx[None, None, :]
    if np.abs(approx - p).max() > rtol * peak:
        return None
    return kz.astype(np.float32), ky.astype(np.float32), kx.astype(np.float32)


def factor_psf_y(psf: np.ndarray, rtol: float = 1e-6):
    """``(ky, kzx)`` with ``psf[z, y, x] == ky[y] * kzx[z, x]`` to within ``rtol * max|psf|``, or ``None``.

    The PSF of an oblique light sheet is tilted in the (z, x) plane and Gaussian along y: it does not
    factor into three 1-D kernels, but it does factor into a y kernel and a dense (z, x) stencil.
    ``ky`` is normalised to sum 1 (the scale stays in ``kzx``).
    """
    p = np.asarray(psf, dtype=np.float64)
    pz, py, px = p.shape
    peak = np.abs(p).max()
    if peak == 0 or py == 1:
        return None
    m = p.transpose(1, 0, 2).reshape(py, pz * px)
    u, sv, vt = np.linalg.svd(m, full_matrices=False)
    ky, kzx = u[:, 0] * sv[0], vt[0].reshape(pz, px)
    if ky.sum() < 0:
        ky, kzx = -ky, -kzx
    sy = ky.sum()
    if sy == 0:
        return None
    ky, kzx = ky / sy, kzx * sy
    if np.abs(ky[None, :, None] * kzx[:, None, :] - p).max() > rtol * peak:
        return None
    return ky.astype(np.float32), kzx.astype(np.float32)


def _axis_norm(k: np.ndarray, n: int) -> np.ndarray:
    """sum of the taps of a 1-D correlation kernel that land inside [0, n), per position."""
    p = len(k)
    c = p // 2
    cs = np.concatenate([[0.0], np.cumsum(k.astype(np.float64))])
    pos = np.arange(n)
    lo = np.maximum(0, c - pos)
    hi = np.minimum(p, n - pos + c)
    return (cs[hi] - cs[lo]).astype(np.float32)


def _prefix_table(w: np.ndarray) -> np.ndarray:
    pz, py, px = w.shape
    t = np.zeros((pz + 1, py + 1, px + 1), dtype=np.float64)
    t[1:, 1:, 1:] = w.astype(np.float64).cumsum(0).cumsum(1).cumsum(2)
    return t


def prepared_dense_taps(psf: np.ndarray):
    """``(taps, taps_flipped)`` host arrays in the tuned dense kernel's layout, or ``None`` when the
    PSF is outside its range (pz <= 11, py, px <= 9): see ``lsr_dense_prepare_taps``."""
    pz, py, px = (int(v) for v in psf.shape)
    lib = _lib.load()
    n = lib.lsr_dense_taps_count(pz, py, px)
    if n < 0:
        return None
    src = np.ascontiguousarray(psf, dtype=np.float32)
    out = []
    for flip in (0, 1):
        buf = np.empty(n, dtype=np.float32)
        _lib.call("lsr_dense_prepare_taps", src.ctypes.data, pz, py, px, flip, buf.ctypes.data)
        out.append(buf)
    return tuple(out)


def padded_shape(shape_zyx, psf_shape):
    """Padded plane geometry the tuned separable kernel reads through (``lsr_sep_padded_shape``).

    Returns ``(rows, pitch, origin_row, origin_col)``: a padded volume is ``(Z, rows, pitch)``
    float32 with the logical ``(Y, X)`` window at ``[origin_row:, origin_col:]`` and zeros elsewhere.
    """
    out = (ctypes.c_int64 * 4)()
    _lib.call("lsr_sep_padded_shape", int(shape_zyx[1]), int(shape_zyx[2]), int(psf_shape[0]),
              int(psf_shape[1]), int(psf_shape[2]), out)
    return tuple(int(v) for v in out)


class PaddedVolume:
    """A zero-haloed working volume: ``.full`` is the allocation, ``.view`` the logical window."""

    def __init__(self, shape_zyx, psf_shape, device):
        import torch

        z, y, x = (int(v) for v in shape_zyx)
        rows, pitch, oy, ox = padded_shape(shape_zyx, psf_shape)
        self.full = torch.zeros((z, rows, pitch), dtype=torch.float32, device=device)
        self.view = self.full[:, oy:oy + y, ox:ox + x]
        self.pitch, self.plane = pitch, rows * pitch

    def logical_ptr(self) -> int:
        return self.view.data_ptr()


@dataclass
class RLStats:
    """Per-iteration reduction scalars of a Richardson-Lucy run (``include/lsrecon.h``, ``lsr_rl_*_stats_f32``), summed
    by the kernels in the epilogue that writes the new estimate -- no extra pass over the volume.

    ``flux[i]   = sum x_i * H^T(ratio_i) = sum x_{i+1} * H^T 1`` (what the update conserves; ``-> sum y`` as ``eps -> 0``),
    ``change[i] = sum |x_{i+1} - x_i|``, ``total[i] = sum x_{i+1}``; ``rel_change = change / total`` is what ``tol`` tests.
    ``iterations`` = launches that ran (``< `` the requested count when ``tol`` stopped the loop)."""

    flux: np.ndarray
    change: np.ndarray
    total: np.ndarray
    iterations: int
    stopped_by_tol: bool = False

    @property
    def rel_change(self) -> np.ndarray:
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.where(self.total > 0, self.change / self.total, 0.0)

    @classmethod
    def from_array(cls, a, iterations: int, stopped: bool = False) -> "RLStats":
        a = np.asarray(a, dtype=np.float64).reshape(-1, 3)[:iterations]
        return cls(flux=a[:, 0].copy(), change=a[:, 1].copy(), total=a[:, 2].copy(), iterations=int(iterations),
                   stopped_by_tol=bool(stopped))


@dataclass
class _DevicePsf:
    separable: bool
    shape: tuple[int, int, int]
    # separable
    k: tuple | None = None
    k_flipped: tuple | None = None
    # dense
    w: object | None = None
    w_flipped: object | None = None
    norm_table: object | None = None
    taps: object | None = None          # tuned dense kernel: host-prepared tap arrays (device)
    taps_flipped: object | None = None
    norm_full: float = 1.0


def fused_pays(pz: int, py: int, px: int) -> bool:
    """Whether the one-launch iteration beats the ratio / update pair for a separable PSF of this extent.

    Its tile shrinks from 32 x 128 to 16 x 128 as the in-plane extent grows (the window's halo takes the LDS), and
    with 13 or 15 in-plane taps and a short z extent the pair is ~10 % faster; everywhere else the fused launch wins
    by 20-40 % (config-2 grid, ``tools/bench_kernels.py --psf-sweep --psf-sweep-wide``,
    ``profiles/r03_rl_psf_sweep.jsonl``: e.g. 5x13x13 4.79 vs 4.28 ms, 11x13x13 5.14 vs 5.34, 9x7x7 2.50 vs 4.31)."""
    return max(py, px) <= 11 or pz >= 11


class RichardsonLucyPlan:
    """PSF taps, border normalisation and scratch for one (volume shape, PSF, device).

    ``plan(y)`` runs ``iterations`` RL iterations and returns the estimate (a new tensor).
    """

    def __init__(self, shape_zyx, psf, device, *, separable: str = "auto",
                 separable_rtol: float = 1e-6, psf_factors=None, fused: str = "auto",
                 y_window: tuple[int, int] | None = None):
        """``y_window = (first_row, total_rows)``: this plan's volume is the row slab
        ``[first_row, first_row + shape_zyx[1])`` of a taller volume of ``total_rows`` rows; the
        border normalisation along y is then the taller volume's (``shrimpy_amd.slab``)."""
        import torch

        if fused not in ("auto", "never", "always"):
            raise ValueError("fused must be 'auto', 'never' or 'always'")
        self._fused_mode = fused
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.LsrError("RichardsonLucyPlan", -1,
                                f"device {self.device} is not a GPU: the plan owns padded DEVICE volumes; CPU tensors run the host "
                                "twins through richardson_lucy() / shrimpy_amd.host")
        self.shape = tuple(int(v) for v in shape_zyx)
        if len(self.shape) != 3 or min(self.shape) <= 0:
            raise ValueError(f"shape_zyx must be three positive ints, got {self.shape}")
        if separable not in ("auto", "force", "never"):
            raise ValueError("separable must be 'auto', 'force' or 'never'")

        factors = None
        if psf_factors is not None:
            factors = tuple(np.asarray(k, dtype=np.float32).ravel() for k in psf_factors)
            if (len(factors) != 3 or any(len(k) % 2 == 0 for k in factors) or len(factors[0]) > MAX_Z_TAPS
                    or max(len(factors[1]), len(factors[2])) > MAX_TAPS):
                raise ValueError("psf_factors must be three odd-length 1-D kernels (<= 31 taps along z, <= 15 in plane)")
            self.psf = (factors[0][:, None, None] * factors[1][None, :, None]
                        * factors[2][None, None, :]).astype(np.float32)
        else:
            self.psf = prepare_psf(psf)
            if separable != "never":
                factors = factor_psf(self.psf, separable_rtol)
                if factors is None and separable == "force":
                    raise ValueError("psf is not rank-1 within separable_rtol")
            if factors is None and self.psf.shape[0] > MAX_TAPS:
                raise ValueError(f"a PSF with {self.psf.shape[0]} z taps must be separable (kz x ky x kx within "
                                 f"separable_rtol): the dense and ky (x) kzx stencils hold <= {MAX_TAPS} taps per axis")

        def dev(a, dtype=torch.float32):
            # (np.array copies: a reversed 1-element view keeps its negative stride otherwise)
            return torch.as_tensor(np.array(a, order="C"), device=self.device).to(dtype)

        z, y, x = self.shape
        # an axial factor beyond the tiled kernels' 15 taps: every correlation = in-plane launch + z launch
        self._long_z = factors is not None and len(factors[0]) > MAX_TAPS
        self._t_dense = None
        if factors is not None:
            kz, ky, kx = factors
            self._psf = _DevicePsf(
                separable=True,
                shape=(len(kz), len(ky), len(kx)),
                k=(dev(kz), dev(ky), dev(kx)),
                k_flipped=(dev(kz[::-1]), dev(ky[::-1]), dev(kx[::-1])),
            )
            ny = _axis_norm(ky, y)
            if y_window is not None:
                first, total = (int(v) for v in y_window)
                if first < 0 or first + y > total:
                    raise ValueError(f"y_window {y_window} does not contain {y} rows")
                ny = _axis_norm(ky, total)[first:first + y]
            self._norm = (dev(_axis_norm(kz, z)), dev(ny), dev(_axis_norm(kx, x)))
            self._one = dev(np.ones(1, np.float32))
        else:
            w = self.psf
            ysep = factor_psf_y(w, separable_rtol) if separable == "auto" and y_window is None else None
            self._ysep = None
            if ysep is not None and len(ysep[0]) <= MAX_TAPS:
                ky, kzx = ysep
                zx = np.ascontiguousarray(kzx[:, None, :])                     # a (pz, 1, px) PSF
                zx_taps = prepared_dense_taps(zx)
                if zx_taps is not None:                                        # pz <= 11, px <= 9
                    one = dev(np.ones(1, np.float32))
                    emb = np.zeros(w.shape, np.float32)                        # kzx in the centre y row
                    emb[:, w.shape[1] // 2, :] = kzx
                    fused_taps = prepared_dense_taps(emb)                      # also needs py <= 9
                    self._ysep = dict(
                        fused=None if fused_taps is None else dict(
                            taps=dev(fused_taps[0]), taps_flipped=dev(fused_taps[1]),
                            norm_table=dev(_prefix_table(ky[None, :, None] * kzx[:, None, :]).ravel(), torch.float64),
                            norm_full=float((ky.astype(np.float64)[None, :, None]
                                             * kzx.astype(np.float64)[:, None, :]).sum())),
                        ky=dev(ky), ky_flipped=dev(ky[::-1]), one=one, py=len(ky), zx_shape=zx.shape,
                        taps=dev(zx_taps[0]), taps_flipped=dev(zx_taps[1]),
                        norm_table=dev(_prefix_table(zx).ravel(), torch.float64),
                        norm_full=float(zx.astype(np.float64).sum()),
                        ny=dev(_axis_norm(ky, y)), ones_z=dev(np.ones(z, np.float32)),
                        ones_x=dev(np.ones(x, np.float32)))
            self._psf = _DevicePsf(
                separable=False,
                shape=tuple(w.shape),
                w=dev(w.ravel()),
                w_flipped=dev(w[::-1, ::-1, ::-1].ravel()),
                norm_table=dev(_prefix_table(w).ravel(), torch.float64),
                norm_full=float(w.astype(np.float64).sum()),
            )
            taps = prepared_dense_taps(w) if self._ysep is None else None
            if taps is not None:  # PSF small enough for the tuned dense kernel
                self._psf.taps, self._psf.taps_flipped = dev(taps[0]), dev(taps[1])
            self._norm = None
        if factors is not None:
            self._ysep = None
        # ky (x) kzx with both factors inside the kernels' range: one launch per ITERATION (rl_fused_ysep.hip; results
        # bit-identical to the pair) -- since round 4 the faster form at every compiled extent (config-2 grid, ms per
        # iteration, one launch / pair: 3x3x3 2.27 / 4.71, 5x5x5 2.82 / 4.57, 9x7x7 4.37 / 5.02, 9x9x9 5.14 / 5.46,
        # 11x9x9 6.19 / 6.95; profiles/r04_ysep_sweep.jsonl).  fused="never" keeps one launch per correlation.
        self.fused_ysep = False
        if (self._ysep is not None and self._ysep["fused"] is not None and self._fused_mode in ("auto", "always")
                and _lib.call_value("lsr_rl_ysep_fused_supported", *self._psf.shape)):
            ky, kzx = ysep
            ky_c, kzx_c = np.ascontiguousarray(ky, dtype=np.float32), np.ascontiguousarray(kzx, dtype=np.float32)
            block = np.zeros(_lib.call_value("lsr_rl_ysep_fused_taps_count"), np.float32)
            _lib.call("lsr_rl_ysep_fused_prepare_taps", ky_c.ctypes.data, len(ky_c), kzx_c.ctypes.data,
                      kzx_c.shape[0], kzx_c.shape[1], block.ctypes.data)
            self._ysep["iter_taps"] = dev(block)
            self.fused_ysep = True
        self._t_pad = None   # y-separable path: the intermediate between the (z, x) and the y pass
        self._ratio = None   # dense path: ratio scratch
        self._x_pad = None   # separable path: zero-haloed working volumes
        self._ratio_pad = None
        self._y_pad = None   # fused path: padded copy of a dense y
        # one launch per iteration (rl_fused_sep.hip) where the PSF fits its specialisations
        self.fused = bool(self._psf.separable and not self._long_z and self._fused_mode in ("auto", "always")
                          and _lib.call_value("lsr_rl_sep_fused_supported", *self._psf.shape)
                          and (self._fused_mode == "always" or fused_pays(*self._psf.shape)))
        if self.fused:
            kz, ky, kx = (np.ascontiguousarray(k, dtype=np.float32) for k in factors)
            block = np.zeros(_lib.call_value("lsr_rl_sep_fused_taps_count"), np.float32)
            _lib.call("lsr_rl_sep_fused_prepare_taps", kz.ctypes.data, len(kz), ky.ctypes.data, len(ky),
                      kx.ctypes.data, len(kx), block.ctypes.data)
            self._fused_taps = dev(block)

    @property
    def separable(self) -> bool:
        return self._psf.separable

    @property
    def padded_input(self) -> bool:
        """``plan(y)`` can take a zero-haloed :class:`PaddedVolume` written in place by the producer of ``y``."""
        return self.path != "generic"

    @property
    def path(self) -> str:
        """Which kernels an iteration runs: ``fused`` (one launch), ``separable`` (ratio / update
        pair), ``y-separable`` ((z, x) stencil + y pass, twice), ``dense`` or ``generic``."""
        if self._psf.separable:
            if self._long_z:
                return "separable (long z, 4 launches)"
            return "fused" if self.fused else "separable"
        if self._ysep is not None:
            if self.fused_ysep:
                return "y-separable (fused)"
            return "y-separable" if self._ysep["fused"] is not None else "y-separable (4 launches)"
        return "dense" if self._psf.taps is not None else "generic"

    def _scratch(self):
        import torch

        if self._psf.separable or self._psf.taps is not None or self._ysep is not None:
            if self._x_pad is None:
                self._x_pad = PaddedVolume(self.shape, self._pad_psf_shape(), self.device)
                self._ratio_pad = PaddedVolume(self.shape, self._pad_psf_shape(), self.device)
            return self._x_pad, self._ratio_pad
        if self._ratio is None:
            self._ratio = torch.empty(self.shape, dtype=torch.float32, device=self.device)
        return self._ratio

    def padded_geometry(self):
        """(pitch, plane, rows, origin_row, origin_col) of this plan's padded working volumes."""
        rows, pitch, oy, ox = padded_shape(self.shape, self._pad_psf_shape())
        return pitch, rows * pitch, rows, oy, ox

    def new_padded_input(self) -> "PaddedVolume":
        """A zero-haloed volume in this plan's geometry, for a producer (the deskew kernel) to write
        ``y`` into; pass it to ``plan(...)`` to skip both the pad copy and the ``x0 = y`` copy."""
        return PaddedVolume(self.shape, self._pad_psf_shape(), self.device)

    def _pad_psf_shape(self):
        """The PSF extents the padded working volumes are laid out for (the z extent of a long-z PSF is not the tiled
        kernels' business: they run with one z tap)."""
        return (1,) + tuple(self._psf.shape[1:]) if self._long_z else tuple(self._psf.shape)

    def _iterate_long_z(self, y_ptr, y_pitch, y_plane, init, x_out, it0, n, eps, stream, stats):
        """RL for a separable PSF with 17 .. 31 z taps: H x = Cz(Cyx(x)) -- the in-plane factors through the tiled
        separable kernel with a single z tap, the z factor through ``lsr_correlate_z_f32`` (a register march, no halo),
        which also carries the epilogues.  Four launches and 48 algorithmic bytes per voxel and iteration."""
        import torch

        x_pad, ratio_pad = self._scratch()
        if it0 == 0:
            x_pad.view.copy_(init)
        if self._t_dense is None:
            self._t_dense = torch.empty(self.shape, dtype=torch.float32, device=self.device)
        t = self._t_dense
        z, yy, xx = self.shape
        pitch, plane = x_pad.pitch, x_pad.plane
        (kz, ky, kx), (fz, fy, fx) = self._psf.k, self._psf.k_flipped
        nz, ny, nx = self._norm
        pz, py, px = self._psf.shape
        one = self._one
        ceps, f0 = ctypes.c_float(eps), ctypes.c_float(0.0)
        for it in range(it0, it0 + n):
            last = x_out is not None and it + 1 == it0 + n
            for src, wy, wx, wz, epi, aux, out in (
                    (x_pad, fy, fx, fz, _lib.EPI_RATIO, (y_ptr, y_pitch, y_plane), (ratio_pad.logical_ptr(), pitch, plane)),
                    (ratio_pad, ky, kx, kz, _lib.EPI_UPDATE, (x_pad.logical_ptr(), pitch, plane),
                     (x_out.data_ptr(), xx, yy * xx) if last else (x_pad.logical_ptr(), pitch, plane))):
                _lib.call("lsr_correlate_sep_strided_f32", src.logical_ptr(), pitch, plane, None, 0, 0, t.data_ptr(), xx,
                          yy * xx, z, yy, xx, one.data_ptr(), 1, wy.data_ptr(), py, wx.data_ptr(), px, _lib.EPI_NONE, f0,
                          None, None, None, stream)
                _lib.call("lsr_correlate_z_f32", t.data_ptr(), xx, yy * xx, aux[0], aux[1], aux[2], out[0], out[1], out[2],
                          z, yy, xx, wz.data_ptr(), pz, epi, ceps, nz.data_ptr(), ny.data_ptr(), nx.data_ptr(),
                          None if (stats is None or epi != _lib.EPI_UPDATE) else stats.data_ptr() + 24 * it, stream)

    def iterate_padded(self, y_pad: "PaddedVolume", src: "PaddedVolume", dst: "PaddedVolume",
                       eps: float = 1e-6, stats=None) -> None:
        """One fused RL iteration between padded volumes of this plan's geometry: reads ``src``
        (and ``y_pad``), writes the logical window of ``dst``; no copies, no allocation.  The
        building block of the slab split (``shrimpy_amd.slab``), where halo rows are refreshed
        between iterations.  ``stats``: a float64 device tensor of 3 elements that receives this
        iteration's (flux, change, total) over THIS volume (a slab's share: sum over the slabs)."""
        import torch

        if not self.fused:
            raise _lib.LsrUnsupported("iterate_padded", _lib.E_UNSUPPORTED,
                                      "needs the fused separable path (PSF within its specialisations)")
        geo = self.padded_geometry()[:2]
        for v, name in ((y_pad, "y_pad"), (src, "src"), (dst, "dst")):
            if (v.pitch, v.plane) != geo or tuple(v.view.shape) != self.shape:
                raise ValueError(f"{name} does not have this plan's padded geometry")
        if src is dst:
            raise ValueError("src and dst must be different volumes")
        z, yy, xx = self.shape
        ps = self._psf
        nz, ny, nx = self._norm
        with torch.cuda.device(self.device):
            # x_a = src (iteration 0 reads it), x_b = dst (iteration 0 writes it)
            if stats is not None and (stats.dtype != torch.float64 or stats.numel() < 3 or stats.device != self.device):
                raise ValueError("stats must be a float64 tensor of >= 3 elements on the plan's device")
            _lib.call(
                "lsr_rl_sep_fused_stats_f32", y_pad.logical_ptr(), y_pad.pitch, y_pad.plane, 0,
                src.full.data_ptr(), dst.full.data_ptr(), None, z, yy, xx, self._fused_taps.data_ptr(),
                ps.shape[0], ps.shape[1], ps.shape[2], nz.data_ptr(), ny.data_ptr(), nx.data_ptr(), 1,
                ctypes.c_float(eps), None if stats is None else stats.data_ptr(), _lib.stream_ptr(self.device),
            )

    def _ysep_nzx(self):
        """(Z, X) table of the (z, x) stencil's border normalisation (sum of the kzx taps that land inside), float64 on
        the device -- only the four-launch path's flux needs it (its last launch divides by ny alone)."""
        import torch

        q = self._ysep
        if "nzx" not in q:
            z, _, x = self.shape
            kzx = self.psf.astype(np.float64).sum(axis=1)          # ky sums to 1: the (z, x) stencil
            pz, px = kzx.shape
            cs = np.zeros((pz + 1, px + 1))
            cs[1:, 1:] = kzx.cumsum(0).cumsum(1)
            zi, xi = np.arange(z), np.arange(x)
            a0, a1 = np.maximum(0, pz // 2 - zi), np.minimum(pz, z - zi + pz // 2)
            c0, c1 = np.maximum(0, px // 2 - xi), np.minimum(px, x - xi + px // 2)
            t = cs[a1][:, c1] - cs[a0][:, c1] - cs[a1][:, c0] + cs[a0][:, c0]
            q["nzx"] = torch.as_tensor(t, device=self.device)
        return q["nzx"]

    def _iterate_ysep(self, y_ptr, y_pitch, y_plane, init, x_out, it0, n, eps, stream, stats):
        """RL with ``psf = ky (x) kzx``, iterations ``it0 .. it0 + n - 1`` (the estimate lives in ``x_pad`` between
        calls; ``init`` is copied in when ``it0 == 0``): each correlation is the dense (z, x) stencil (PZ * PX FMAs per
        voxel instead of PZ * PY * PX) followed by the y pass that carries the epilogue --
        ``H x = Y~(ZX~(x))`` with ``ratio = y / (. + eps)`` in the y pass; ``H^T r = Y(ZX(r))`` with the
        (z, x) border normalisation in the stencil launch (``LSR_EPI_SCALE``) and ``x * . / ny`` in
        the y pass.  Four launches per iteration (two where both factors fit one kernel), all on zero-haloed padded
        volumes.  ``stats``: float64 device tensor (iterations, 3), zeroed by the caller, or ``None``."""
        import torch

        q = self._ysep
        x_pad, ratio_pad = self._scratch()
        if it0 == 0:
            x_pad.view.copy_(init)
        z, yy, xx = self.shape
        pitch, plane = x_pad.pitch, x_pad.plane
        ceps = ctypes.c_float(eps)

        def sptr(it):
            return None if stats is None else stats.data_ptr() + 24 * it

        if q["fused"] is not None:
            # both factors in one launch per correlation (lsr_correlate_zxy_padded_f32): the y pass
            # runs inside the stencil kernel, wave by wave, on the staged plane
            f = q["fused"]
            pz, py, px = self._psf.shape
            for it in range(it0, it0 + n):
                last = x_out is not None and it + 1 == it0 + n
                _lib.call("lsr_correlate_zxy_padded_f32", x_pad.logical_ptr(), pitch, plane, y_ptr, y_pitch, y_plane,
                          ratio_pad.logical_ptr(), pitch, plane, z, yy, xx, f["taps_flipped"].data_ptr(),
                          q["ky_flipped"].data_ptr(), pz, py, px, _lib.EPI_RATIO, ceps, None,
                          ctypes.c_float(1.0), stream)
                out_ptr, out_pitch, out_plane = ((x_out.data_ptr(), xx, yy * xx) if last
                                                 else (x_pad.logical_ptr(), pitch, plane))
                _lib.call("lsr_correlate_zxy_padded_stats_f32", ratio_pad.logical_ptr(), pitch, plane, x_pad.logical_ptr(),
                          pitch, plane, out_ptr, out_pitch, out_plane, z, yy, xx, f["taps"].data_ptr(),
                          q["ky"].data_ptr(), pz, py, px, _lib.EPI_UPDATE, ceps, f["norm_table"].data_ptr(),
                          ctypes.c_float(f["norm_full"]), sptr(it), stream)
            return
        if self._t_pad is None:
            self._t_pad = PaddedVolume(self.shape, self._psf.shape, self.device)
        t_pad = self._t_pad
        pz, _, px = q["zx_shape"]
        one, f0 = q["one"].data_ptr(), ctypes.c_float(0.0)
        for it in range(it0, it0 + n):
            last = x_out is not None and it + 1 == it0 + n
            _lib.call("lsr_correlate_dense_padded_f32", x_pad.logical_ptr(), pitch, plane, None, 0, 0,
                      t_pad.logical_ptr(), pitch, plane, z, yy, xx, q["taps_flipped"].data_ptr(), pz, 1, px,
                      _lib.EPI_NONE, f0, None, ctypes.c_float(1.0), stream)
            _lib.call("lsr_correlate_sep_strided_f32", t_pad.logical_ptr(), pitch, plane, y_ptr, y_pitch, y_plane,
                      ratio_pad.logical_ptr(), pitch, plane, z, yy, xx, one, 1, q["ky_flipped"].data_ptr(), q["py"],
                      one, 1, _lib.EPI_RATIO, ceps, None, None, None, stream)
            _lib.call("lsr_correlate_dense_padded_f32", ratio_pad.logical_ptr(), pitch, plane, None, 0, 0,
                      t_pad.logical_ptr(), pitch, plane, z, yy, xx, q["taps"].data_ptr(), pz, 1, px,
                      _lib.EPI_SCALE, f0, q["norm_table"].data_ptr(), ctypes.c_float(q["norm_full"]), stream)
            out_ptr, out_pitch, out_plane = ((x_out.data_ptr(), xx, yy * xx) if last
                                             else (x_pad.logical_ptr(), pitch, plane))
            _lib.call("lsr_correlate_sep_strided_stats_f32", t_pad.logical_ptr(), pitch, plane, x_pad.logical_ptr(), pitch,
                      plane, out_ptr, out_pitch, out_plane, z, yy, xx, one, 1, q["ky"].data_ptr(), q["py"], one, 1,
                      _lib.EPI_UPDATE, ceps, q["ones_z"].data_ptr(), q["ny"].data_ptr(), q["ones_x"].data_ptr(),
                      sptr(it), stream)
            if stats is not None:
                # this launch's x * u is x * H^T(ratio) / nzx (the (z, x) normalisation went into the stencil launch): the
                # flux of the full update, sum x_new * nzx * ny, is formed here from the new estimate -- this path is the
                # fallback for y extents beyond the one-launch kernels, an extra reduction does not matter to it
                x_new = x_out if last else x_pad.view
                rows = torch.einsum("zyx,y->zx", x_new, q["ny"])
                stats[it, 0] = (rows.double() * self._ysep_nzx()).sum()

    def release(self) -> None:
        """Drop the scratch volumes."""
        self._ratio = self._x_pad = self._ratio_pad = self._y_pad = self._t_pad = self._t_dense = None

    # ------------------------------------------------------------------------------------------ the loop
    def _launch(self, st, it0: int, n: int, x_out, stats) -> None:
        """Iterations ``it0 .. it0 + n - 1`` of the run described by ``st`` (set up by ``__call__``); the last one
        writes the dense ``x_out`` when that is not ``None``, otherwise the estimate stays in the working volume
        ``_result_view(it0 + n)`` names.  ``stats``: float64 device tensor (iterations, 3), zero where not yet run."""
        z, yy, xx = self.shape
        ps = self._psf
        stream, eps = st["stream"], ctypes.c_float(st["eps"])
        y_ptr, y_pitch, y_plane = st["y"]
        from_y = int(st["from_y"] and it0 == 0)
        xo = None if x_out is None else x_out.data_ptr()
        sp = None if stats is None else stats.data_ptr() + 24 * it0
        kind = st["kind"]
        if kind in ("fused", "fused-ysep"):
            x_pad, ratio_pad = self._scratch()
            bufs = (x_pad.full.data_ptr(), ratio_pad.full.data_ptr())   # iteration i reads bufs[i & 1], writes the other
            a, b = bufs[it0 & 1], bufs[(it0 + 1) & 1]
            if kind == "fused":
                nz, ny, nx = self._norm
                _lib.call("lsr_rl_sep_fused_stats_f32", y_ptr, y_pitch, y_plane, from_y, a, b, xo, z, yy, xx,
                          self._fused_taps.data_ptr(), ps.shape[0], ps.shape[1], ps.shape[2], nz.data_ptr(),
                          ny.data_ptr(), nx.data_ptr(), n, eps, sp, stream)
            else:
                f = self._ysep["fused"]
                _lib.call("lsr_rl_ysep_fused_stats_f32", y_ptr, y_pitch, y_plane, from_y, a, b, xo, z, yy, xx,
                          self._ysep["iter_taps"].data_ptr(), ps.shape[0], ps.shape[1], ps.shape[2],
                          f["norm_table"].data_ptr(), ctypes.c_float(f["norm_full"]), n, eps, sp, stream)
        elif kind == "separable":
            x_pad, ratio_pad = self._scratch()
            (kz, ky, kx), (fz, fy, fx) = ps.k, ps.k_flipped
            nz, ny, nx = self._norm
            _lib.call("lsr_rl_sep_stats_f32", y_ptr, y_pitch, y_plane, from_y, x_pad.full.data_ptr(),
                      ratio_pad.full.data_ptr(), xo, z, yy, xx, kz.data_ptr(), fz.data_ptr(), ps.shape[0],
                      ky.data_ptr(), fy.data_ptr(), ps.shape[1], kx.data_ptr(), fx.data_ptr(), ps.shape[2],
                      nz.data_ptr(), ny.data_ptr(), nx.data_ptr(), n, eps, sp, stream)
        elif kind == "ysep":
            self._iterate_ysep(y_ptr, y_pitch, y_plane, st["init"], x_out, it0, n, st["eps"], stream, stats)
        elif kind == "longz":
            self._iterate_long_z(y_ptr, y_pitch, y_plane, st["init"], x_out, it0, n, st["eps"], stream, stats)
        elif kind == "dense":
            x_pad, ratio_pad = self._scratch()
            _lib.call("lsr_rl_dense_padded_stats_f32", y_ptr, y_pitch, y_plane, from_y, x_pad.full.data_ptr(),
                      ratio_pad.full.data_ptr(), xo, z, yy, xx, ps.taps.data_ptr(), ps.taps_flipped.data_ptr(),
                      ps.shape[0], ps.shape[1], ps.shape[2], ps.norm_table.data_ptr(), ctypes.c_float(ps.norm_full),
                      n, eps, sp, stream)
        else:   # generic: dense volumes, x updated in place
            _lib.call("lsr_rl_dense_stats_f32", st["y_dense"].data_ptr(), st["x"].data_ptr(), self._scratch().data_ptr(),
                      z, yy, xx, ps.w.data_ptr(), ps.w_flipped.data_ptr(), ps.shape[0], ps.shape[1], ps.shape[2],
                      ps.norm_table.data_ptr(), n, eps, sp, stream)

    def _result_view(self, st, done: int):
        """Where the estimate is after ``done`` iterations that did not write the dense output."""
        if st["kind"] == "generic":
            return st["x"]
        x_pad, ratio_pad = self._scratch()
        if st["kind"] in ("fused", "fused-ysep"):
            return (x_pad, ratio_pad)[done & 1].view
        return x_pad.view

    def __call__(self, y, iterations: int = 20, eps: float = 1e-6, x0=None, out=None, events=None, *,
                 stats: bool = False, tol: float | None = None):
        """Run RL.  ``events`` = optional ``(start, end)`` torch events recorded on the launch
        stream right around the kernel launches (``iterations`` fused launches, or
        ``2 * iterations`` ratio / update launches) -- what ``bench.py`` times.

        ``stats=True``: the kernels also sum the iteration's reduction scalars in their epilogues
        (:class:`RLStats`; read them afterwards from ``plan.last_stats`` -- the device tensor is
        ``plan.stats_device``, one row per iteration).  ``tol``: stop as soon as the relative change
        ``sum|x_new - x| / sum x_new`` of an iteration falls below it.  The scalars of iteration i are read back while
        iteration i + 1 runs, so the GPU never waits for the host; the estimate returned is therefore the one
        iteration past the first that met ``tol`` (``plan.last_stats.iterations`` says how many ran)."""
        import torch

        y_padded = None
        if isinstance(y, PaddedVolume):  # e.g. written in place by the deskew kernel
            if not (self._psf.separable or self._psf.taps is not None or self._ysep is not None):
                y = y.view.contiguous()
            else:
                y_padded, y = y, y.view
        else:
            y = _lib.require_device_f32(y, "y")
        if tuple(y.shape) != self.shape or y.device != self.device:
            raise ValueError(f"y must be {self.shape} on {self.device}, got {tuple(y.shape)} on {y.device}")
        if y_padded is not None and (y_padded.pitch, y_padded.plane) != self.padded_geometry()[:2]:
            raise ValueError("the padded y does not have this plan's padded geometry")
        iterations = int(iterations)
        if iterations < 0:
            raise ValueError("iterations must be >= 0")
        if not eps > 0:
            raise ValueError("eps must be > 0")
        if tol is not None and not (tol >= 0 and np.isfinite(tol)):
            raise ValueError("tol must be a finite number >= 0")
        # x0 = y with a padded y needs no initial copy: the first iteration reads y directly
        from_y = y_padded is not None and x0 is None and iterations > 0
        y_ptr, y_pitch, y_plane = ((y_padded.logical_ptr(), y_padded.pitch, y_padded.plane)
                                   if y_padded is not None else (y.data_ptr(), self.shape[2],
                                                                 self.shape[1] * self.shape[2]))
        init = y if x0 is None else _lib.require_device_f32(x0, "x0")
        if tuple(init.shape) != self.shape:
            raise ValueError(f"x0 must be {self.shape}")
        if out is None:
            x = torch.empty(self.shape, dtype=torch.float32, device=self.device)
        else:
            x = _lib.require_device_f32(out, "out")
            if tuple(x.shape) != self.shape or x.data_ptr() == y.data_ptr():
                raise ValueError("out must have the volume shape and must not alias y")
        self.last_stats = None
        self.stats_device = None
        if iterations == 0:
            x.copy_(init)
            if stats or tol is not None:
                self.last_stats = RLStats.from_array(np.zeros((0, 3)), 0)
            return x
        ps = self._psf
        want_stats = bool(stats) or tol is not None
        with torch.cuda.device(self.device):
            st = dict(stream=_lib.stream_ptr(self.device), eps=float(eps), init=init)
            padded_y_kinds = {"fused": ps.separable and self.fused,
                              "fused-ysep": (not ps.separable) and self._ysep is not None and self.fused_ysep}
            if padded_y_kinds["fused"] or padded_y_kinds["fused-ysep"]:
                # one launch per iteration; y must be a zero-haloed padded volume (the kernel
                # reads it on the tile grown by the PSF radius)
                st["kind"] = "fused" if padded_y_kinds["fused"] else "fused-ysep"
                x_pad, _ = self._scratch()
                if y_padded is None:
                    if self._y_pad is None:
                        self._y_pad = PaddedVolume(self.shape, ps.shape, self.device)
                    self._y_pad.view.copy_(y)
                    y_padded = self._y_pad
                    y_ptr, y_pitch, y_plane = y_padded.logical_ptr(), y_padded.pitch, y_padded.plane
                    from_y = x0 is None
                if not from_y:
                    x_pad.view.copy_(init)
            elif self._long_z:
                st["kind"] = "longz"
            elif ps.separable or ps.taps is not None and self._ysep is None:
                # working volumes carry a zero halo: the kernels never bounds-check a load
                st["kind"] = "separable" if ps.separable else "dense"
                x_pad, _ = self._scratch()
                if not from_y:
                    x_pad.view.copy_(init)
            elif self._ysep is not None:
                st["kind"] = "ysep"
            else:
                st["kind"] = "generic"
                x.copy_(init)
                st["x"], st["y_dense"] = x, y
            st["y"], st["from_y"] = (y_ptr, y_pitch, y_plane), from_y
            dev_stats = None
            if want_stats:
                dev_stats = torch.zeros((iterations, 3), dtype=torch.float64, device=self.device)
            if events:
                events[0].record()
            if tol is None:
                self._launch(st, 0, iterations, None if st["kind"] == "generic" else x, dev_stats)
                done, stopped = iterations, False
            else:
                done, stopped = self._run_to_tolerance(st, iterations, float(tol), dev_stats)
                if st["kind"] != "generic":
                    x.copy_(self._result_view(st, done))
            if events:
                events[1].record()
            if want_stats:
                self.stats_device = dev_stats
                self.last_stats = RLStats.from_array(dev_stats.cpu().numpy(), done, stopped)
        _lib.mark_written(x)
        return x

    def _run_to_tolerance(self, st, iterations: int, tol: float, dev_stats):
        """One launch group per iteration; iteration i's scalars travel to pinned host memory behind it and are looked at
        after iteration i + 1 has been queued.  Returns (iterations run, stopped early)."""
        import torch

        host = torch.empty((iterations, 3), dtype=torch.float64).pin_memory()
        arrived = [torch.cuda.Event() for _ in range(iterations)]

        def met(i):
            arrived[i].synchronize()
            change, total = float(host[i, 1]), float(host[i, 2])
            return total > 0 and change <= tol * total or total == 0

        done = 0
        for it in range(iterations):
            self._launch(st, it, 1, None, dev_stats)
            host[it].copy_(dev_stats[it], non_blocking=True)
            arrived[it].record()
            done = it + 1
            if it >= 1 and met(it - 1):
                return done, True
        return done, bool(met(iterations - 1))


def make_plan(shape_zyx, psf, device, *, separable: str = "auto", separable_rtol: float = 1e-6, psf_factors=None,
              fused: str = "auto", method: str = "auto"):
    """The plan that runs RL for this volume shape and PSF on a HIP device.

    ``method="direct"``: the stencil kernels (:class:`RichardsonLucyPlan`; PSFs up to 15 taps per axis, 31 along z
    when separable).  ``"fft"``: the two convolutions of an iteration as products of spectra
    (:class:`shrimpy_amd.deconvolve_fft.FftRichardsonLucyPlan`; any PSF the transform grid holds).  ``"auto"``: the
    stencil kernels wherever a tuned one takes the PSF (three 1-D factors, ``ky (x) kzx``, dense up to 11 x 9 x 9);
    a dense PSF beyond them -- a measured bead PSF -- goes to the Fourier domain, where its size costs nothing
    (``profiles/r04_rl_fft.jsonl``), and only falls back to the bounds-checked generic stencil when the grid is
    outside the transform kernels' lengths."""
    if method not in ("auto", "direct", "fft"):
        raise ValueError("method must be 'auto', 'direct' or 'fft'")
    from .deconvolve_fft import MAX_FFT_TAPS, FftRichardsonLucyPlan, fft_supported

    def dense_psf():
        if psf_factors is not None:
            kz, ky, kx = (np.asarray(k, dtype=np.float32).ravel() for k in psf_factors)
            return (kz[:, None, None] * ky[None, :, None] * kx[None, None, :]).astype(np.float32)
        return prepare_psf(psf, MAX_FFT_TAPS, MAX_FFT_TAPS)

    if method == "fft":
        return FftRichardsonLucyPlan(shape_zyx, dense_psf(), device)
    direct, refused = None, None
    try:
        direct = RichardsonLucyPlan(shape_zyx, psf, device, separable=separable, separable_rtol=separable_rtol,
                                    psf_factors=psf_factors, fused=fused)
    except ValueError as exc:      # more taps than the stencil kernels hold
        if method == "direct" or psf_factors is not None or psf is None:
            raise
        refused = exc
    # (the tuned dense stencil costs one FMA per tap: at 11 x 9 x 9 = 891 taps it is the slower route -- 26.9 against
    # 23.1 ms per iteration on the config-2 grid, 15.2 against 23.0 ms at 9 x 7 x 7; profiles/r04_rl_fft.jsonl)
    if method == "direct" or (direct is not None and direct.path != "generic"
                              and not (direct.path == "dense" and int(np.prod(direct.psf.shape)) > 800)):
        return direct
    w = dense_psf()
    if fft_supported(shape_zyx, w.shape):
        return FftRichardsonLucyPlan(shape_zyx, w, device)
    if direct is None:
        raise refused
    return direct


def richardson_lucy(y, psf=None, iterations: int = 20, eps: float = 1e-6, x0=None, *,
                    separable: str = "auto", separable_rtol: float = 1e-6, psf_factors=None,
                    tol: float | None = None, return_stats: bool = False, method: str = "auto"):
    """Richardson-Lucy deconvolution of a (Z, Y, X) float32 device tensor; returns a new tensor.

    ``psf`` is used as given (normalise it to sum 1 for flux conservation).  ``x0`` defaults to
    ``y``.  ``separable="auto"`` takes the rank-1 fast path when the PSF factorises within
    ``separable_rtol``; pass ``psf_factors=(kz, ky, kx)`` to skip the test.

    ``method``: ``"auto"`` (default) runs the stencil kernels where a tuned one takes the PSF and the Fourier-domain
    iteration for dense PSFs beyond them (:func:`make_plan`); ``"direct"`` / ``"fft"`` insist on one.  CPU tensors
    always run the host twins of the stencil arithmetic.

    ``tol``: stop before ``iterations`` once an iteration's relative change ``sum|x_new - x| / sum x_new`` is below
    it (the kernels sum both in their epilogues; see :class:`RLStats`).  ``return_stats=True`` returns
    ``(estimate, RLStats)`` -- flux, change and total per iteration that ran.
    """
    import torch

    if not isinstance(y, torch.Tensor):
        raise TypeError(f"y must be a torch.Tensor, got {type(y).__name__}")
    if y.dim() != 3:
        raise ValueError(f"y must be (Z, Y, X), got shape {tuple(y.shape)}")
    if y.dtype != torch.float32:
        y = y.to(torch.float32)
    y = y.contiguous()
    if y.device.type == "cpu":     # no HIP device in play: the host twins of the two launches per iteration
        from . import host

        return host.richardson_lucy(y, psf, iterations, eps, x0, separable=separable, separable_rtol=separable_rtol,
                                    psf_factors=psf_factors, tol=tol, return_stats=return_stats)
    plan = make_plan(tuple(y.shape), psf, y.device, separable=separable, separable_rtol=separable_rtol,
                     psf_factors=psf_factors, method=method)
    x = plan(y, iterations=iterations, eps=eps, x0=x0, stats=return_stats, tol=tol)
    return (x, plan.last_stats) if return_stats else x


def correlate3d(volume, weights=None, *, weight_factors=None, tuned: bool = True):
    """``scipy.ndimage.correlate(volume, weights, mode="constant", cval=0)`` on the device.

    Pass ``weight_factors=(wz, wy, wx)`` for the separable kernel.  ``tuned=False`` forces the
    generic bounds-checked dense kernel (any size up to 15 taps per axis).  (The un-fused building
    block of the RL launches; also used for PSF-blurring synthetic scenes.)
    """
    import torch

    if isinstance(volume, torch.Tensor) and volume.device.type == "cpu":
        from . import host

        return host.correlate3d(volume, weights, weight_factors)
    vol = _lib.require_device_f32(volume, "volume")
    if vol.dim() != 3:
        raise ValueError("volume must be (Z, Y, X)")
    out = torch.empty_like(vol)
    z, y, x = (int(v) for v in vol.shape)

    def dev(a):
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device=vol.device)

    with torch.cuda.device(vol.device):
        stream = _lib.stream_ptr(vol.device)
        if weight_factors is not None:
            wz, wy, wx = (np.asarray(k, dtype=np.float32).ravel() for k in weight_factors)
            dz, dy, dx = dev(wz), dev(wy), dev(wx)
            pad = PaddedVolume(vol.shape, (len(wz), len(wy), len(wx)), vol.device)
            pad.view.copy_(vol)
            _lib.call(
                "lsr_correlate_sep_strided_f32", pad.logical_ptr(), pad.pitch, pad.plane, None, 0, 0,
                out.data_ptr(), x, y * x, z, y, x, dz.data_ptr(), len(wz), dy.data_ptr(), len(wy),
                dx.data_ptr(), len(wx), _lib.EPI_NONE, ctypes.c_float(0.0), None, None, None, stream,
            )
        else:
            w = prepare_psf(weights)
            taps = prepared_dense_taps(w) if tuned else None
            if taps is not None:
                dt = dev(taps[0])
                pad = PaddedVolume(vol.shape, w.shape, vol.device)
                pad.view.copy_(vol)
                _lib.call(
                    "lsr_correlate_dense_padded_f32", pad.logical_ptr(), pad.pitch, pad.plane, None, 0, 0,
                    out.data_ptr(), x, y * x, z, y, x, dt.data_ptr(), w.shape[0], w.shape[1], w.shape[2],
                    _lib.EPI_NONE, ctypes.c_float(0.0), None, ctypes.c_float(1.0), stream,
                )
            else:
                dw = dev(w.ravel())
                _lib.call(
                    "lsr_correlate_dense_f32", vol.data_ptr(), out.data_ptr(), None, z, y, x,
                    dw.data_ptr(), w.shape[0], w.shape[1], w.shape[2], _lib.EPI_NONE,
                    ctypes.c_float(0.0), None, stream,
                )
        # the tap tensors must outlive the launch: the caching allocator keeps their blocks on
        # this stream, so reuse after free is stream-ordered
    return out
