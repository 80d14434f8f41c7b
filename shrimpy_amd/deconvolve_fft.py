"""Richardson-Lucy deconvolution with the two convolutions of an iteration done in the Fourier domain.

The stencil kernels of :mod:`shrimpy_amd.deconvolve` hold PSFs of up to 15 taps per axis (31 along z when the
PSF is separable), and a dense PSF costs them one FMA per tap and voxel.  A *measured* PSF is neither small nor
separable: the bead patches the PSF-characterisation script around the reference averages are 15 x 18 x 18 to
30 x 36 x 18 voxels (``/root/reference/scripts/measure_psf.py:187-190``).  For those the iteration

    x <- x * H^T( y / (H x + eps) ) / H^T 1          (zero-padded borders)

runs here with ``H x`` and ``H^T r`` as products of spectra on a grid of at least ``volume + PSF radius`` points per
axis (a linear convolution: what wraps around lands in the padding), at a cost that does not depend on the PSF's
size.  There is no reference code for RL (``/root/reference/docs/data_structure.md:58-62``); the arithmetic is that
of the stencil path, and the test oracle is ``oracle.cpu_ref.richardson_lucy`` (direct and ``use_fft=True``).

One convolution = five launches, none of which writes a real-space volume other than the iteration's own ``ratio`` and
``x``.  (``LSR_FFT_RL_CHAIN=1`` runs steps 5 and 1 of consecutive convolutions as ONE kernel, ``lsr_rl_rows_chain_f32``:
eight launches per iteration, the ratio never in memory, the same bits -- and 23.5 instead of 23.0 ms per iteration on
the config-2 grid: the row kernels are bound by the latency chain load -> transform -> epilogue inside a workgroup, two
of which fit a CU, not by the 6.4 GB the chaining saves.  Kept as the form to start from once the transforms are faster.)

1. ``lsr_rfft_rows_zero_t_c64``    zero padding + real-to-complex transform along x + transpose (LDS-resident rows)
2. hipFFT, batched 1-D, in place   the long y axis (the one leg left to the library, as in :mod:`shrimpy_amd.fft3`)
3. ``lsr_spectrum_multiply_z_c64`` forward z transform, product with the PSF's spectrum (or its conjugate), inverse z
4. hipFFT                          y back
5. ``lsr_irfft_rows_rl_f32``       complex-to-real transform along x, crop, and the iteration's epilogue: the ratio
                                   ``y / (H x + eps)`` or the update ``x * H^T r / H^T 1`` with the reduction scalars

Grid lengths: z 5-smooth and <= 256, x a multiple of 4 whose half is 5-smooth and <= 2048, y 5-smooth -- i.e. volumes
up to ``(256 - pz // 2, any, 4096 - px // 2)``, which covers every deskewed stack of BASELINE.json's configs.
"""

from __future__ import annotations

import ctypes

import numpy as np

from . import _lib, fft3

__all__ = ["FftRichardsonLucyPlan", "fft_grid", "fft_supported"]

MAX_FFT_TAPS = 129     # per axis; beyond that the padding outgrows the volume for any stack this package sees


def _next_smooth(n: int) -> int:
    """Smallest 5-smooth integer >= n."""
    n = max(int(n), 1)
    while True:
        m = n
        for f in (2, 3, 5):
            while m % f == 0:
                m //= f
        if m == 1:
            return n
        n += 1


def fft_grid(shape_zyx, psf_shape) -> tuple[int, int, int]:
    """The transform grid of a volume and a PSF: per axis ``max(n + p // 2, p)`` (the volume plus the PSF radius the
    wrap-around must land in; at least the PSF itself for volumes thinner than it) rounded up to a length the kernels
    take."""
    (z, y, x), (pz, py, px) = (tuple(int(v) for v in shape_zyx), tuple(int(v) for v in psf_shape))
    nz, ny, nx = max(z + pz // 2, pz), max(y + py // 2, py), max(x + px // 2, px)
    # (the z leg transforms at least 2 points, the x leg at least 8)
    return (_next_smooth(max(nz, 2)), _next_smooth(ny), 4 * _next_smooth(max(-(-nx // 4), 2)))


def fft_supported(shape_zyx, psf_shape) -> bool:
    """The x and z legs' own kernels take this grid and hipFFT's C API is loadable for the y leg."""
    if max(int(v) for v in psf_shape) > MAX_FFT_TAPS:
        return False
    gz, _, gx = fft_grid(shape_zyx, psf_shape)
    return fft3.rows_supported((gz, 8, gx))


class FftRichardsonLucyPlan:
    """Spectrum of the PSF, border normalisation and scratch for one (volume shape, PSF, device).

    Same call contract as :class:`shrimpy_amd.deconvolve.RichardsonLucyPlan`: ``plan(y)`` runs the iterations and
    returns the estimate (a new tensor, or ``out``); ``stats=True`` / ``tol=`` as there.  (The float32 transforms leave
    ~1e-7 of every voxel as rounding: the relative change ``sum|x_new - x| / sum x_new`` bottoms out near 1e-7 instead of
    reaching 0 at a fixed point, so a ``tol`` below ~1e-6 never stops this route early.)"""

    padded_input = False     # y is a plain dense volume (the stencil plans take zero-haloed ones)
    path = "fft"
    separable = False
    fused = False

    def __init__(self, shape_zyx, psf, device):
        import torch

        from .deconvolve import _prefix_table, prepare_psf

        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.LsrError("FftRichardsonLucyPlan", -1, f"device {self.device} is not a GPU: CPU tensors run the host "
                                "twins of the stencil path through richardson_lucy()")
        self.shape = tuple(int(v) for v in shape_zyx)
        if len(self.shape) != 3 or min(self.shape) <= 0:
            raise ValueError(f"shape_zyx must be three positive ints, got {self.shape}")
        self.psf = prepare_psf(psf, MAX_FFT_TAPS, MAX_FFT_TAPS)
        if not fft_supported(self.shape, self.psf.shape):
            raise _lib.LsrError("FftRichardsonLucyPlan", _lib.E_UNSUPPORTED,
                                f"volume {self.shape} with a {self.psf.shape} PSF: the transform grid must be <= 256 "
                                "along z and <= 4096 along x (and hipFFT loadable for the y leg)")
        self.grid = fft_grid(self.shape, self.psf.shape)
        gz, gy, gx = self.grid
        self._xc = gx // 2 + 1
        w = self.psf.astype(np.float64)
        self._norm_full = float(w.sum())
        if not self._norm_full > 0:
            raise ValueError("the PSF must have a positive sum")
        with torch.cuda.device(self.device):
            self._norm_table = torch.as_tensor(_prefix_table(self.psf).ravel(), device=self.device).to(torch.float64)
            # the PSF on the grid with its centre tap at the origin (wrapped), then its spectrum [XC][Y][Z]
            pz, py, px = self.psf.shape
            g = torch.zeros(self.grid, dtype=torch.float32, device=self.device)
            g[:pz, :py, :px] = torch.as_tensor(self.psf, device=self.device)
            g = torch.roll(g, shifts=(-(pz // 2), -(py // 2), -(px // 2)), dims=(0, 1, 2)).contiguous()
            self._otf = fft3.spectrum_of(g, self.grid)
            del g
        self._half, self._full = fft3._row_twiddles(gx, self.device)
        self._tw_z = fft3._twiddle_table(gz, self.device)
        self._scale = 1.0 / (float(gz) * float(gy) * float(gx))
        self._b = None         # [Z][XC][Y] complex64: the one spectrum in flight
        self._ratio = None
        self.last_stats = None
        self.stats_device = None

    def release(self) -> None:
        self._b = self._ratio = None

    def _scratch(self):
        import torch

        import os

        if self._b is None:
            gz, gy, _ = self.grid
            self._b = torch.empty((gz, self._xc, gy), dtype=torch.complex64, device=self.device)
        if self._ratio is None and os.environ.get("LSR_FFT_RL_CHAIN", "0") != "1":
            self._ratio = torch.empty(self.shape, dtype=torch.float32, device=self.device)
        return self._b, self._ratio

    def _forward(self, src) -> None:
        """``b`` <- the x leg of ``src``'s spectrum (zero padding + transform + transpose)."""
        gz, gy, gx = self.grid
        z, y, x = self.shape
        b, _ = self._scratch()
        _lib.call("lsr_rfft_rows_zero_t_c64", src.data_ptr(), z, y, x, b.data_ptr(), gz, gy, gx, self._half.data_ptr(),
                  self._full.data_ptr(), _lib.stream_ptr(self.device))

    def _middle(self, conj: int) -> None:
        """``b`` (x leg done) <- y transform, z transform x PSF spectrum (or its conjugate) x inverse z, y back."""
        gz, gy, _ = self.grid
        z = self.shape[0]
        b, _ = self._scratch()
        dev = self.device
        # only the volume's z planes exist before the z transform (the padding behind them is zeros: z_valid) and
        # only they are wanted after it (z_keep): the y legs run on those planes alone
        fft3._exec(dev, fft3._HIPFFT_C2C, gy, z * self._xc, b.data_ptr(), b.data_ptr(), fft3._FORWARD)
        _lib.call("lsr_spectrum_multiply_z_c64", self._otf.data_ptr(), b.data_ptr(), self._tw_z.data_ptr(), gz, gy,
                  self._xc, conj, z, z, _lib.stream_ptr(dev))
        fft3._exec(dev, fft3._HIPFFT_C2C, gy, z * self._xc, b.data_ptr(), b.data_ptr(), fft3._BACKWARD)

    def _epilogue(self, entry: str, epilogue: int, aux, out, eps: float, stats_ptr) -> None:
        gz, gy, gx = self.grid
        z, y, x = self.shape
        pz, py, px = self.psf.shape
        b, _ = self._scratch()
        _lib.call(entry, b.data_ptr(), gz, gy, gx, self._half.data_ptr(), self._full.data_ptr(), epilogue,
                  aux.data_ptr(), None if out is None else out.data_ptr(), z, y, x, ctypes.c_float(self._scale),
                  ctypes.c_float(eps), pz, py, px, self._norm_table.data_ptr(), ctypes.c_float(self._norm_full), stats_ptr,
                  _lib.stream_ptr(self.device))

    def _iteration(self, x, y, eps: float, stats_row, first: bool) -> None:
        """``LSR_FFT_RL_CHAIN=1``: one iteration in eight launches.  ``b`` holds the x leg of the current estimate's spectrum on entry (``first``:
        made here) and of the new estimate's on exit: both inverse x legs are chained into the next forward one
        (``lsr_rl_rows_chain_f32``), the ratio is never written, x_new is not read back for its transform."""
        if first:
            self._forward(x)
        self._middle(0)
        self._epilogue("lsr_rl_rows_chain_f32", _lib.EPI_RATIO, y, None, eps, None)            # b <- x leg of y / (H x + eps)
        self._middle(1)
        self._epilogue("lsr_rl_rows_chain_f32", _lib.EPI_UPDATE, x, x, eps,                     # x <- x H^T r / H^T 1
                       None if stats_row is None else stats_row.data_ptr())

    def _iteration_unchained(self, x, y, eps: float, stats_row) -> None:
        """The iteration as ten launches with the ratio volume in memory: the default (see the module docstring)."""
        _, ratio = self._scratch()
        self._forward(x)
        self._middle(0)
        self._epilogue("lsr_irfft_rows_rl_f32", _lib.EPI_RATIO, y, ratio, eps, None)
        self._forward(ratio)
        self._middle(1)
        self._epilogue("lsr_irfft_rows_rl_f32", _lib.EPI_UPDATE, x, x, eps, None if stats_row is None else stats_row.data_ptr())

    def _step(self, x, y, eps, stats_row, first):
        import os

        if os.environ.get("LSR_FFT_RL_CHAIN", "0") != "1":
            self._iteration_unchained(x, y, eps, stats_row)
        else:
            self._iteration(x, y, eps, stats_row, first)

    def __call__(self, y, iterations: int = 20, eps: float = 1e-6, x0=None, out=None, events=None, *,
                 stats: bool = False, tol: float | None = None):
        """Run RL; arguments as :meth:`shrimpy_amd.deconvolve.RichardsonLucyPlan.__call__`.  ``events``: ``(start, end)``
        torch events recorded right around the launches (ten per iteration)."""
        import torch

        from .deconvolve import RLStats

        y = _lib.require_device_f32(y, "y")
        if tuple(y.shape) != self.shape or y.device != self.device:
            raise ValueError(f"y must be {self.shape} on {self.device}, got {tuple(y.shape)} on {y.device}")
        iterations = int(iterations)
        if iterations < 0:
            raise ValueError("iterations must be >= 0")
        if not eps > 0:
            raise ValueError("eps must be > 0")
        if tol is not None and not (tol >= 0 and np.isfinite(tol)):
            raise ValueError("tol must be a finite number >= 0")
        if out is None:
            out = torch.empty(self.shape, dtype=torch.float32, device=self.device)
        elif (tuple(out.shape) != self.shape or out.dtype != torch.float32 or out.device != self.device
              or not out.is_contiguous()):
            raise ValueError(f"out must be a contiguous float32 {self.shape} tensor on {self.device}")
        if out.data_ptr() == y.data_ptr():
            raise ValueError("out must not alias y (y is read by every iteration)")
        init = y if x0 is None else _lib.require_device_f32(x0, "x0")
        if tuple(init.shape) != self.shape:
            raise ValueError(f"x0 must be {self.shape}, got {tuple(init.shape)}")
        want_stats = bool(stats) or tol is not None
        with torch.cuda.device(self.device):
            if out.data_ptr() != init.data_ptr():
                out.copy_(init)
            dev_stats = torch.zeros((iterations, 3), dtype=torch.float64, device=self.device) if want_stats else None
            if events:
                events[0].record()
            done, stopped = iterations, False
            if tol is None:
                for it in range(iterations):
                    self._step(out, y, float(eps), None if dev_stats is None else dev_stats[it], it == 0)
            else:
                done, stopped = self._run_to_tolerance(out, y, float(eps), iterations, float(tol), dev_stats)
            if events:
                events[1].record()
            if want_stats:
                self.stats_device = dev_stats
                self.last_stats = RLStats.from_array(dev_stats.cpu().numpy(), done, stopped)
        _lib.mark_written(out)
        return out

    def _run_to_tolerance(self, x, y, eps, iterations, tol, dev_stats):
        """Iteration i's scalars travel to pinned host memory behind it and are looked at after iteration i + 1 has
        been queued (as the stencil plans do): the estimate returned is the one past the first that met ``tol``."""
        import torch

        if iterations == 0:
            return 0, False
        host = torch.empty((iterations, 3), dtype=torch.float64).pin_memory()
        arrived = [torch.cuda.Event() for _ in range(iterations)]

        def met(i):
            arrived[i].synchronize()
            change, total = float(host[i, 1]), float(host[i, 2])
            return total > 0 and change <= tol * total or total == 0

        done = 0
        for it in range(iterations):
            self._step(x, y, eps, dev_stats[it], it == 0)
            host[it].copy_(dev_stats[it], non_blocking=True)
            arrived[it].record()
            done = it + 1
            if it >= 1 and met(it - 1):
                return done, True
        return done, bool(met(iterations - 1))
