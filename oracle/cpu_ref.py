"""CPU oracle for the light-sheet reconstruction hot path.  TEST INFRASTRUCTURE ONLY.

This file is the *checker*, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
``shrimpy_amd`` never imports anything under ``oracle/``.

PARITY UNPINNED.  The arithmetic of this path is not in /root/reference: deskew lives in
the un-vendored dependency ``biahub`` (0.0.1rc2.post17 @ b011bca57d5bf15c777839e4250e594f7471af3e,
reference ``pyproject.toml:91``, ``uv.lock:323-325``); affine registration and
Richardson-Lucy have no code or call site in the reference at all
(``docs/data_structure.md:58-62``), and every reference test that touches deskew stubs it out
(``shrimpy/tests/test_preprocessing.py:12-13``).  There are therefore no golden vectors to pin
against.  What this oracle follows instead:

* the ``scipy.ndimage`` path that ``BASELINE.json:north_star`` names as the CPU reference
  (``affine_transform(order=1, mode="constant", cval=0)``; explicit RL loop over
  ``ndimage.convolve``),
* the published biahub geometry (``biahub/deskew.py`` at the pinned revision: output->input
  matrix, ``Z_shift`` flooring, ``ceil`` of the scan extent, edge-padded slice averaging),
* the reference's own call sites, which fix names, argument meaning and axis conventions:
  ``shrimpy/preprocessing.py:226-231, 408-413`` (keyword call, kwargs filtered by signature),
  ``scripts/measure_psf.py:223-249`` (ratio rounding to 3 decimals; chunks split on raw X are
  concatenated *reversed* on output axis -2; voxel size used as a 3-tuple scale),
  ``shrimpy/viewer/ring_buffer.py:98-105`` (one deskewed plane = one tilt row across the scan
  stack), ``config/mda/mantis/dynatrack_demo.yaml:161-164`` (defaults 30 deg,
  keep_overhang false, average 3).

Array layout everywhere: numpy C-order ``(Z, Y, X)``; raw stacks are
``(Z=scan, Y=tilt, X=coverslip)`` (``scripts/measure_psf.py:91,101``).
"""

from __future__ import annotations

import math

import numpy as np
from scipy import ndimage, signal

# --------------------------------------------------------------------------------------
# Deskew geometry  (biahub.deskew.get_deskewed_data_shape / deskew_data; call sites
# shrimpy/preprocessing.py:226-231, scripts/measure_psf.py:230-246)
# --------------------------------------------------------------------------------------


def deskew_geometry(raw_shape, ls_angle_deg, px_to_scan_ratio, keep_overhang):
    """Return ``(matrix3x3, offset3, pre_average_shape)`` of the output->input affine map.

    Rows (scipy convention, output index -> input coordinate)::

        z_in = -r*cos(t) * Z' + r * X' + Z_shift
        y_in = -Z' + (Y - 1)
        x_in = -Y' + (X - 1)

    so Z' walks the tilt rows reversed, Y' walks raw X reversed (1:1, no interpolation) and
    X' is the scan direction -- the only interpolated axis.
    """
    Z, Y, X = (int(v) for v in raw_shape)
    ct = math.cos(ls_angle_deg * math.pi / 180.0)
    r = float(px_to_scan_ratio)
    if keep_overhang:
        z_shift = 0
        xp = int(math.ceil(Z / r + Y * ct))
    else:
        z_shift = int(math.floor(Y * ct * r))
        xp = int(math.ceil(Z / r - Y * ct))
    matrix = np.array([[-r * ct, 0.0, r], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]], dtype=np.float64)
    offset = np.array([float(z_shift), float(Y - 1), float(X - 1)], dtype=np.float64)
    return matrix, offset, (Y, X, xp)


def deskewed_shape(
    raw_shape, ls_angle_deg, px_to_scan_ratio, keep_overhang, average_n_slices=1, pixel_size_um=1.0
):
    """Output shape and voxel size (restates biahub ``get_deskewed_data_shape``)."""
    _, _, (zd, yd, xd) = deskew_geometry(raw_shape, ls_angle_deg, px_to_scan_ratio, keep_overhang)
    st = math.sin(ls_angle_deg * math.pi / 180.0)
    shape = (int(math.ceil(zd / average_n_slices)), yd, xd)
    voxel = (average_n_slices * st * pixel_size_um, pixel_size_um, pixel_size_um)
    return shape, voxel


def average_slices(data, n):
    """Mean over groups of ``n`` slices along axis 0; the remainder is edge-padded.

    Restates biahub ``_average_n_slices`` (pad ``mode="edge"``, reshape, mean) in float32.
    """
    n = int(n)
    if n <= 1:
        return data
    rem = data.shape[0] % n
    if rem:
        data = np.pad(data, [(0, n - rem), (0, 0), (0, 0)], mode="edge")
    grouped = data.reshape((data.shape[0] // n, n) + data.shape[1:])
    return grouped.mean(axis=1, dtype=np.float32)


def affine_apply(volume, matrix, offset, output_shape, cval=0.0, mode="constant"):
    """``scipy.ndimage.affine_transform`` order-1, the north-star's CPU path.

    ``mode="constant"``: a sample with any coordinate outside ``[0, n-1]`` is ``cval`` (no
    blending).  ``mode="grid-constant"``: blends towards ``cval`` across the border.
    """
    volume = np.ascontiguousarray(volume, dtype=np.float32)
    return ndimage.affine_transform(
        volume,
        np.asarray(matrix, dtype=np.float64),
        offset=np.asarray(offset, dtype=np.float64),
        output_shape=tuple(int(s) for s in output_shape),
        order=1,
        mode=mode,
        cval=float(cval),
        prefilter=False,
    )


def affine_apply_4x4(volume, matrix_4x4, output_shape, cval=0.0, mode="constant"):
    """Registration apply: 4x4 ZYX homogeneous matrix mapping target -> source voxel coords."""
    m = np.asarray(matrix_4x4, dtype=np.float64)
    return affine_apply(volume, m[:3, :3], m[:3, 3], output_shape, cval=cval, mode=mode)


def orient(volume, orientation="identity"):
    """Re-orient a canonical deskewed volume: ``+``-joined numpy operations applied left to right
    (``flip_z|flip_y|flip_x``, ``transpose_yx``, ``rot90|rot180|rot270`` = ``np.rot90(axes=(1, 2))``).
    The orientation ``fast_deskew_zyx`` returns is a [RECALLED] convention (SURVEY.md section 8 a2,
    stale comment ``shrimpy/preprocessing.py:224``), hence a switch."""
    for op in str(orientation).split("+"):
        op = op.strip()
        if op == "identity":
            continue
        if op in ("flip_z", "flip_y", "flip_x"):
            volume = np.flip(volume, axis="zyx".index(op[-1]))
        elif op == "transpose_yx":
            volume = np.swapaxes(volume, 1, 2)
        elif op in ("rot90", "rot180", "rot270"):
            volume = np.rot90(volume, k=int(op[3:]) // 90, axes=(1, 2))
        else:
            raise ValueError(f"unknown orientation op {op!r}")
    return np.ascontiguousarray(volume)


def deskew(raw, ls_angle_deg, px_to_scan_ratio, keep_overhang, average_n_slices=1,
           orientation="identity", border="constant", cval=0.0):
    """Deskew a raw ``(Z_scan, Y_tilt, X)`` stack -> ``(ceil(Y/avg), X, Xp)`` float32.

    ``border`` is scipy's ``mode`` (``"constant"``: no blending, the north-star's path;
    ``"grid-constant"``: blend towards zero, the torch ``grid_sample`` behaviour).  ``cval``: scipy's fill value, or
    ``"min"`` / ``None`` for the stack's minimum ([RECALLED] biahub ``deskew_data``'s rule when its cval is None)."""
    if cval is None or isinstance(cval, str):
        cval = float(np.asarray(raw).min())
    # the fill value is a float32 value, like every sample of the stack it stands beside (the product hands the kernel a
    # float32 scalar): under "grid-constant" scipy blends with it in double precision, and 132.8 and float32(132.8) then
    # give results one ulp apart (found by tests/soak_parity.py, round 5)
    cval = float(np.float32(cval))
    matrix, offset, pre_shape = deskew_geometry(
        raw.shape, ls_angle_deg, px_to_scan_ratio, keep_overhang
    )
    if pre_shape[2] <= 0:
        raise ValueError(f"deskewed scan extent is not positive: {pre_shape}")
    out = affine_apply(raw, matrix, offset, pre_shape, cval=cval, mode=border)
    return orient(average_slices(out, average_n_slices), orientation)


# --------------------------------------------------------------------------------------
# Richardson-Lucy  (north-star: 3-D PSF stencil, zero-padded borders, 20 iterations)
# --------------------------------------------------------------------------------------


def _as_odd_psf(psf):
    """Zero-pad even-sized axes by one trailing plane so that every axis is odd.

    With ``center = size // 2`` the padded kernel gives the same ``ndimage.correlate`` result.
    """
    psf = np.asarray(psf, dtype=np.float32)
    pad = [(0, 1 - (s % 2)) for s in psf.shape]
    if any(p[1] for p in pad):
        psf = np.pad(psf, pad)
    return psf


def rl_norm(shape, psf):
    """``H^T 1``: the fraction of PSF mass that lands inside the volume, per voxel."""
    psf = _as_odd_psf(psf)
    ones = np.ones(shape, dtype=np.float32)
    return ndimage.correlate(ones, psf, mode="constant", cval=0.0)


def richardson_lucy(y, psf, iterations=20, eps=1e-6, x0=None, use_fft=False):
    """Richardson-Lucy with zero-padded borders::

        x <- x * H^T( y / (H x + eps) ) / (H^T 1)

    ``H x = ndimage.convolve(x, psf, mode="constant")`` (direct stencil, float32 in/out,
    double accumulate inside scipy); ``H^T r = ndimage.correlate(r, psf)``.  ``x0 = y`` unless
    given.  The PSF is used as passed (callers normalise it to sum 1).
    """
    y = np.ascontiguousarray(y, dtype=np.float32)
    psf = _as_odd_psf(psf)
    eps32 = np.float32(eps)
    norm = rl_norm(y.shape, psf)
    x = y.copy() if x0 is None else np.ascontiguousarray(x0, dtype=np.float32).copy()
    psf_flip = psf[::-1, ::-1, ::-1]
    for _ in range(int(iterations)):
        if use_fft:
            blur = signal.fftconvolve(x, psf, mode="same").astype(np.float32)
        else:
            blur = ndimage.convolve(x, psf, mode="constant", cval=0.0)
        ratio = y / (blur + eps32)
        if use_fft:
            corr = signal.fftconvolve(ratio, psf_flip, mode="same").astype(np.float32)
        else:
            corr = ndimage.correlate(ratio, psf, mode="constant", cval=0.0)
        x = x * corr / norm
    return x


def rl_iteration_scalars(y, psf, iterations=20, eps=1e-6, x0=None):
    """Per-iteration reduction scalars of :func:`richardson_lucy`, summed in float64 over the float32 estimates:
    ``flux[i] = sum x_{i+1} * H^T 1`` (conserved by the multiplicative update: ``-> sum y`` as ``eps -> 0``),
    ``change[i] = sum |x_{i+1} - x_i|``, ``total[i] = sum x_{i+1}``.  The product's kernels sum the same three in their
    update epilogues (``include/lsrecon.h``: ``lsr_rl_*_stats_f32``; no reference code: the north-star names
    "wavefront reductions for the ratio / normalisation", ``/root/reference/docs/data_structure.md:58-62``)."""
    y = np.ascontiguousarray(y, dtype=np.float32)
    psf = _as_odd_psf(psf)
    norm = rl_norm(y.shape, psf).astype(np.float64)
    x = y.copy() if x0 is None else np.ascontiguousarray(x0, dtype=np.float32).copy()
    out = {k: np.zeros(int(iterations)) for k in ("flux", "change", "total")}
    for i in range(int(iterations)):
        x_new = richardson_lucy(y, psf, iterations=1, eps=eps, x0=x)
        out["flux"][i] = float((x_new.astype(np.float64) * norm).sum())
        out["change"][i] = float(np.abs(x_new.astype(np.float64) - x.astype(np.float64)).sum())
        out["total"][i] = float(x_new.astype(np.float64).sum())
        x = x_new
    return out


def richardson_lucy_separable(y, factors, iterations=20, eps=1e-6, x0=None):
    """The same RL loop for a rank-1 PSF ``kz x ky x kx`` as three ``ndimage.correlate1d``
    passes per correlation -- the fastest scipy.ndimage formulation, used as the CPU baseline
    (``bench.py``) so that the CPU side exploits separability exactly like the device path."""
    y = np.ascontiguousarray(y, dtype=np.float32)
    ks = [np.asarray(k, dtype=np.float32) for k in factors]
    eps32 = np.float32(eps)

    def corr(v, flip):
        for axis, k in enumerate(ks):
            v = ndimage.correlate1d(v, k[::-1] if flip else k, axis=axis, mode="constant", cval=0.0)
        return v

    norm = corr(np.ones(y.shape, dtype=np.float32), False)
    x = y.copy() if x0 is None else np.ascontiguousarray(x0, dtype=np.float32).copy()
    for _ in range(int(iterations)):
        ratio = y / (corr(x, True) + eps32)  # H x = correlate with the flipped factors
        x = x * corr(ratio, False) / norm
    return x


def rl_iteration_parts(x, y, psf, eps=1e-6):
    """One RL iteration split the way the device path splits it (ratio, then update)."""
    psf = _as_odd_psf(psf)
    blur = ndimage.convolve(np.asarray(x, np.float32), psf, mode="constant", cval=0.0)
    ratio = np.asarray(y, np.float32) / (blur + np.float32(eps))
    corr = ndimage.correlate(ratio, psf, mode="constant", cval=0.0)
    norm = rl_norm(x.shape, psf)
    return ratio, np.asarray(x, np.float32) * corr / norm


# --------------------------------------------------------------------------------------
# Neighbour step: flat-field (shrimpy/preprocessing.py:385-404)
# --------------------------------------------------------------------------------------


def flat_field_bf(volume):
    """Divide out the per-pixel median over Z, preserving its mean (float32).

    Follows ``_LabelfreePreprocessor._flat_field_BF`` (``shrimpy/preprocessing.py:403-404``):
    ``quantile(0.5)`` == ``numpy.median`` (mean of the two middle values for even Z).
    """
    v = np.asarray(volume, dtype=np.float32)
    pattern = np.median(v, axis=0).astype(np.float32)
    return v / pattern * pattern.mean(dtype=np.float32)


# --------------------------------------------------------------------------------------
# Synthetic scene + PSF; shared by tests and bench.py's CPU leg
# --------------------------------------------------------------------------------------


def gaussian_psf(shape=(9, 7, 7), sigma=(2.0, 1.2, 1.2)):
    """Separable anisotropic Gaussian PSF truncated to ``shape``, normalised to sum 1.

    Returns ``(psf3d, (kz, ky, kx))`` with ``psf3d == kz[:,None,None]*ky[None,:,None]*kx``.
    """
    ks = []
    for n, s in zip(shape, sigma):
        c = n // 2
        g = np.exp(-0.5 * ((np.arange(n) - c) / s) ** 2)
        ks.append(g / g.sum())
    kz, ky, kx = ks
    psf = (kz[:, None, None] * ky[None, :, None] * kx[None, None, :]).astype(np.float32)
    return psf, tuple(k.astype(np.float32) for k in ks)


def rotated_psf(shape=(9, 7, 7), sigma=(2.0, 1.2, 1.2), angle_deg=30.0):
    """Non-separable PSF: the Gaussian above rotated about Y (mimics the oblique sheet)."""
    cz, cy, cx = (n // 2 for n in shape)
    z, y, x = np.meshgrid(
        np.arange(shape[0]) - cz, np.arange(shape[1]) - cy, np.arange(shape[2]) - cx, indexing="ij"
    )
    a = math.radians(angle_deg)
    zr = math.cos(a) * z + math.sin(a) * x
    xr = -math.sin(a) * z + math.cos(a) * x
    g = np.exp(-0.5 * ((zr / sigma[0]) ** 2 + (y / sigma[1]) ** 2 + (xr / sigma[2]) ** 2))
    return (g / g.sum()).astype(np.float32)


def bead_scene(shape, seed, psf=None, density=2e-5, background=100.0, psf_factors=None):
    """Sparse beads U(200,4000) on a flat background, PSF-blurred, Poisson noise, float32.
    ``psf_factors`` (kz, ky, kx) blurs with three 1-D passes instead (large scenes)."""
    rng = np.random.default_rng(seed)
    n = int(np.prod(shape))
    vol = np.zeros(shape, dtype=np.float32)
    k = max(1, int(round(density * n)))
    idx = rng.integers(0, n, size=k)
    vol.reshape(-1)[idx] = rng.uniform(200.0, 4000.0, size=k).astype(np.float32) * 30.0
    if psf_factors is not None:
        for axis, k in enumerate(psf_factors):
            vol = ndimage.convolve1d(vol, np.asarray(k, np.float32), axis=axis, mode="constant")
    elif psf is not None:
        vol = ndimage.convolve(vol, np.asarray(psf, np.float32), mode="constant")
    vol = vol + np.float32(background)
    return rng.poisson(vol).astype(np.float32)


# --------------------------------------------------------------------------------------------
# Next row f-4, second half: estimating the label-free <-> light-sheet affine.  No reference code
# exists (docs/data_structure.md:58-62; biahub's estimate-registration is [RECALLED]) -- PARITY
# UNPINNED; this restates the textbook additive Gauss-Newton / Lucas-Kanade step the product's kernel
# implements, in numpy, so that the kernel's sums and the recovered transforms can be checked.
# --------------------------------------------------------------------------------------------


def affine_normal_equations(moving, target, matrix_3x4, gain=1.0, offset=0.0, stride=1, centre=None, scale=None):
    """``(H 14x14, b 14, sse, n)`` of  min sum (gain * M(A x) + offset - T(x))^2  at ``matrix_3x4`` over
    the target grid sampled every ``stride`` voxels; parameters = the matrix rows in centred, scaled
    target coordinates ``((x - centre) / scale, 1)``, then gain, offset."""
    mov = np.asarray(moving, np.float64)
    tgt = np.asarray(target, np.float64)
    m = np.asarray(matrix_3x4, np.float64)[:3]
    shape = tgt.shape
    c = np.asarray(centre if centre is not None else [(n - 1) / 2 for n in shape], np.float64)
    s = float(scale if scale is not None else max(shape) / 2)
    strides = (stride,) * 3 if np.isscalar(stride) else tuple(int(v) for v in stride)
    idx = np.stack(np.meshgrid(*[np.arange(0, n, st, dtype=np.float64) for n, st in zip(shape, strides)], indexing="ij"),
                   -1).reshape(-1, 3)
    coord = idx @ m[:, :3].T + m[:, 3]
    lim = np.array(mov.shape) - 1
    keep = np.all((coord >= 0) & (coord < lim), axis=1)
    idx, coord = idx[keep], coord[keep]
    j = np.floor(coord).astype(int)
    f = coord - j
    v = {(a, b_, c_): mov[j[:, 0] + a, j[:, 1] + b_, j[:, 2] + c_] for a in (0, 1) for b_ in (0, 1) for c_ in (0, 1)}
    fz, fy, fx = f[:, 0], f[:, 1], f[:, 2]
    a00 = v[0, 0, 0] + fx * (v[0, 0, 1] - v[0, 0, 0]); a01 = v[0, 1, 0] + fx * (v[0, 1, 1] - v[0, 1, 0])
    a10 = v[1, 0, 0] + fx * (v[1, 0, 1] - v[1, 0, 0]); a11 = v[1, 1, 0] + fx * (v[1, 1, 1] - v[1, 1, 0])
    b0 = a00 + fy * (a01 - a00); b1 = a10 + fy * (a11 - a10)
    mval = b0 + fz * (b1 - b0)
    gz = b1 - b0
    gy = (a01 - a00) + fz * ((a11 - a10) - (a01 - a00))
    d00, d01 = v[0, 0, 1] - v[0, 0, 0], v[0, 1, 1] - v[0, 1, 0]
    d10, d11 = v[1, 0, 1] - v[1, 0, 0], v[1, 1, 1] - v[1, 1, 0]
    e0 = d00 + fy * (d01 - d00); e1 = d10 + fy * (d11 - d10)
    gx = e0 + fz * (e1 - e0)
    tv = tgt[idx[:, 0].astype(int), idx[:, 1].astype(int), idx[:, 2].astype(int)]
    r = gain * mval + offset - tv
    xt = np.concatenate([(idx - c) / s, np.ones((len(idx), 1))], axis=1)
    jac = np.concatenate([gain * g[:, None] * xt for g in (gz, gy, gx)] + [mval[:, None], np.ones((len(idx), 1))], axis=1)
    return jac.T @ jac, jac.T @ r, float(r @ r), int(len(idx))


def estimate_affine(moving, target, initial=None, model="affine", intensity=True, levels=((2, 1.0), (1, 0.0)),
                    max_iterations=40, tol=2e-3):
    """Gauss-Newton with Levenberg-Marquardt damping on the sums above, coarse to fine; returns
    ``(4x4 target index -> moving coordinate, gain, offset, rms)``."""
    shape = np.asarray(target).shape
    c = np.array([(n - 1) / 2 for n in shape])
    s = max(shape) / 2
    m = np.eye(4)[:3].copy() if initial is None else np.asarray(initial, np.float64)[:3].copy()
    mov, tgt = np.asarray(moving, np.float32), np.asarray(target, np.float32)
    gain, offset = 1.0, 0.0
    if intensity and mov.std() > 0 and tgt.std() > 0:
        gain = float(tgt.std() / mov.std())
        offset = float(tgt.mean() - gain * mov.mean())
    free = np.zeros(14, bool)
    free[[3, 7, 11]] = True
    if model == "affine":
        free[:12] = True
    if intensity:
        free[12:] = True
    corners = np.array([[z, y, x, 1.0] for z in (0, shape[0] - 1) for y in (0, shape[1] - 1) for x in (0, shape[2] - 1)])
    rms = np.nan
    for stride, sigma in levels:
        lm = ndimage.gaussian_filter(mov, sigma, mode="reflect", truncate=4.0) if sigma > 0 else mov
        lt = ndimage.gaussian_filter(tgt, sigma, mode="reflect", truncate=4.0) if sigma > 0 else tgt
        h, b, sse, n = affine_normal_equations(lm, lt, m, gain, offset, stride, c, s)
        lam = 1e-3
        for _ in range(max_iterations):
            hf, bf = h[np.ix_(free, free)], b[free]
            accepted = None
            for _try in range(8):
                delta = np.linalg.solve(hf + lam * np.diag(np.diag(hf)) + 1e-12 * np.eye(hf.shape[0]), -bf)
                full = np.zeros(14)
                full[free] = delta
                q = np.concatenate([m[:, :3] * s, (m[:, :3] @ c + m[:, 3])[:, None]], axis=1) + full[:12].reshape(3, 4)
                m_new = np.concatenate([q[:, :3] / s, (q[:, 3] - (q[:, :3] / s) @ c)[:, None]], axis=1)
                g_new, o_new = gain + full[12], offset + full[13]
                h2, b2, sse2, n2 = affine_normal_equations(lm, lt, m_new, g_new, o_new, stride, c, s)
                if n2 >= 0.5 * n and sse2 / max(n2, 1) <= sse / n * (1 + 1e-12):
                    accepted = (m_new, g_new, o_new, h2, b2, sse2, n2)
                    lam = max(lam / 3, 1e-9)
                    break
                lam *= 10
            if accepted is None:
                break
            moved = float(np.abs(corners @ (accepted[0] - m).T).max())
            m, gain, offset, h, b, sse, n = accepted
            if moved < tol:
                break
        rms = float(np.sqrt(sse / n))
    out = np.eye(4)
    out[:3] = m
    return out, gain, offset, rms


# --------------------------------------------------------------------------------------------
# Next row f-3: DynaTrack shift estimators on the deskewed volume
# (shrimpy/dynatrack/tracking.py:386-707, 759-787).  Pinned by tests/golden/ref_dynatrack.npz,
# which oracle/make_golden.py captured by running the reference's own functions.
# --------------------------------------------------------------------------------------------


def dt_histc(img, nbins, vmin, vmax):
    """``torch.histc``: bin = int((x - min) * nbins / (max - min)) in float32 (``tracking.py:465, 586``)."""
    x = np.asarray(img, np.float32).ravel()
    lo, hi = np.float32(vmin), np.float32(vmax)
    x = x[(x >= lo) & (x <= hi)]
    b = ((x - lo) * np.float32(nbins) / (hi - lo)).astype(np.int64)
    return np.bincount(np.minimum(b, nbins - 1), minlength=nbins).astype(np.float32)


def dt_percentile(img, percentile, nbins=256):
    """``_percentile`` (``tracking.py:572-593``): upper edge of the bin where the CDF reaches p."""
    vmin, vmax = float(np.min(img)), float(np.max(img))
    if vmax <= vmin:
        return vmin
    cdf = np.cumsum(dt_histc(img, nbins, vmin, vmax), dtype=np.float32)
    cdf = cdf / cdf[-1]
    idx = min(int(np.searchsorted(cdf, np.float32(percentile / 100.0), side="left")), nbins - 1)
    return vmin + (idx + 1) * (vmax - vmin) / nbins


def dt_intensity_center_of_mass(img, background=0.0):
    """``_intensity_center_of_mass`` (``tracking.py:596-649``)."""
    w = np.maximum(np.asarray(img, np.float32) - np.float32(background), 0).astype(np.float64)
    total = w.sum()
    if total <= 0:
        return np.array([(s - 1) / 2.0 for s in w.shape], np.float32)
    return np.array([(w.sum(axis=tuple(d for d in range(3) if d != a)) * np.arange(w.shape[a])).sum() / total
                     for a in range(3)], np.float32)


def dt_gaussian_blur_3d(img, sigma):
    """``_gaussian_blur_3d`` (``tracking.py:386-422``): three 1-D passes, reflect padding without
    repeating the edge sample (scipy's "mirror"), radius ``int(4 sigma + 0.5)`` clamped to n - 1."""
    vol = np.asarray(img, np.float32)
    if sigma <= 0:
        return vol
    max_radius = int(4 * sigma + 0.5)
    for axis in range(3):
        r = min(max_radius, vol.shape[axis] - 1)
        x = np.arange(-r, r + 1, dtype=np.float32)
        k = np.exp(np.float32(-0.5) * (x / np.float32(sigma)) ** 2).astype(np.float32)
        k = (k / k.sum(dtype=np.float32)).astype(np.float32)
        vol = ndimage.correlate1d(vol, k, axis=axis, mode="mirror").astype(np.float32)
    return vol


def dt_multiotsu_threshold(img_blur, otsu_component=0, nbins=256):
    """``_multiotsu_threshold`` (``tracking.py:425-501``), float32 like the reference."""
    v = np.asarray(img_blur, np.float32)
    vmin, vmax = float(v.min()), float(v.max())
    if vmin == vmax:
        return vmin
    hist = dt_histc(v, nbins, vmin, vmax)
    hist = (hist / hist.sum(dtype=np.float32)).astype(np.float32)
    step = (np.float32(vmax) - np.float32(vmin)) / np.float32(nbins - 1)
    i = np.arange(nbins, dtype=np.float32)
    centers = np.where(np.arange(nbins) < nbins // 2, np.float32(vmin) + step * i,
                       np.float32(vmax) - step * (np.float32(nbins - 1) - i)).astype(np.float32)
    cw = np.cumsum(hist, dtype=np.float32)
    cm = np.cumsum(hist * centers, dtype=np.float32)
    mu, eps = cm[-1], np.float32(1e-10)
    w0, w1, w2 = cw[:, None], cw[None, :] - cw[:, None], np.float32(1) - cw[None, :]
    m0 = cm[:, None] / np.maximum(w0, eps)
    m1 = (cm[None, :] - cm[:, None]) / np.maximum(w1, eps)
    m2 = (mu - cm[None, :]) / np.maximum(w2, eps)
    sig = w0 * (m0 - mu) ** 2 + w1 * (m1 - mu) ** 2 + w2 * (m2 - mu) ** 2
    b = np.arange(nbins)
    ok = (b[None, :] > b[:, None]) & (b[None, :] <= nbins - 2) & (w0 > eps) & (w1 > eps) & (w2 > eps)
    a_, b_ = divmod(int(np.argmax(np.where(ok, sig, np.float32(-1)))), nbins)
    return (float(centers[a_ + 1]), float(centers[b_ + 1]))[min(otsu_component, 1)]


def dt_binary_mask(img, sigma=5.0, otsu_component=0):
    """``_binary_mask`` (``tracking.py:504-542``)."""
    v = np.asarray(img, np.float32)
    vmin, vmax = v.min(), v.max()
    if not vmax > vmin:
        return np.zeros(v.shape, bool)
    blur = dt_gaussian_blur_3d((v - vmin) / (vmax - vmin), sigma)
    return blur > dt_multiotsu_threshold(blur, otsu_component)


def dt_center_of_mass(mask):
    """``_center_of_mass`` (``tracking.py:545-569``)."""
    c = np.argwhere(mask)
    return c.mean(axis=0).astype(np.float32) if len(c) else np.zeros(mask.ndim, np.float32)


def dt_roi_shift(img, background_percentile=None, blur_sigma=0.0):
    """``_intensity_center_of_mass_to_roi_center`` (``tracking.py:652-707``)."""
    v = np.asarray(img, np.float32)
    if blur_sigma and blur_sigma > 0:
        v = dt_gaussian_blur_3d(v, blur_sigma)
    bg = dt_percentile(v, background_percentile) if background_percentile is not None else 0.0
    return dt_intensity_center_of_mass(v, bg) - np.array([(s - 1) / 2.0 for s in v.shape], np.float32)


def dt_multiotsu_center_of_mass(ref_img, mov_img, sigma=5.0, otsu_component=0):
    """``_multiotsu_center_of_mass`` (``tracking.py:759-787``)."""
    return (dt_center_of_mass(dt_binary_mask(mov_img, sigma, otsu_component))
            - dt_center_of_mass(dt_binary_mask(ref_img, sigma, otsu_component)))


def dt_next_fast_len(n):
    """``_next_fast_len`` (``tracking.py:248-263``)."""
    n = max(int(n), 1)
    while True:
        m = n
        for p in (2, 3, 5):
            while m % p == 0:
                m //= p
        if m == 1:
            return n
        n += 1


def dt_match_shape(t, shape):
    """``_match_shape`` (``tracking.py:266-306``): reflect-pad (left = d // 2), then centre-crop."""
    t = np.asarray(t, np.float32)
    pad = [((max(s - a, 0)) // 2, max(s - a, 0) - max(s - a, 0) // 2) for s, a in zip(shape, t.shape)]
    t = np.pad(t, pad, mode="reflect")
    sl = tuple(slice((a - s) // 2, (a - s) // 2 + s) for s, a in zip(shape, t.shape))
    return t[sl]


def dt_phase_cross_corr(ref, mov, maximum_shift=1.0):
    """``_phase_cross_corr`` (``tracking.py:309-378``)."""
    ref, mov = np.asarray(ref, np.float32), np.asarray(mov, np.float32)
    shape = tuple(dt_next_fast_len(int(max(a, b) * maximum_shift)) for a, b in zip(ref.shape, mov.shape))
    f1 = np.fft.rfftn(dt_match_shape(ref, shape))
    f2 = np.fft.rfftn(dt_match_shape(mov, shape))
    corr = np.fft.fftshift(np.abs(np.fft.irfftn(f1 * np.conj(f2), s=shape, axes=tuple(range(len(shape))))))
    peak = np.unravel_index(int(np.argmax(corr)), corr.shape)
    return tuple(int(s // 2) - int(p) for s, p in zip(corr.shape, peak))


def dt_limit_shifts_zyx(shifts_zyx, limits):
    """``_limit_shifts_zyx`` (``tracking.py:822-868``): deadband below min, clip above max."""
    v = np.array(shifts_zyx, dtype=float)
    for i, axis in enumerate("zyx"):
        if axis in limits:
            lo, hi = limits[axis]
            if abs(v[i]) < lo:
                v[i] = 0.0
            elif abs(v[i]) > hi:
                v[i] = np.sign(v[i]) * hi
    return v


def dt_compute_shift(ref, mov, method, scale_z, scale_yx, maximum=1.0, limits=None, dampening=None,
                     otsu_sigma=5.0, otsu_component=0, blob_sigma=10.0, background_percentile=None, blur_sigma=0.0):
    """``DynaTrackUpdater._compute_shift`` (``tracking.py:1224-1312``) over the ``dt_*`` estimators."""
    if method == "pcc":
        px = dt_phase_cross_corr(ref, mov, maximum)
    elif method == "intensity_center_of_mass":
        px = dt_roi_shift(mov, background_percentile, blur_sigma)
    elif method == "multiotsu_center_of_mass":
        px = dt_multiotsu_center_of_mass(ref, mov, otsu_sigma, otsu_component)
    else:
        raise ValueError(method)
    um = np.array([px[0] * scale_z, px[1] * scale_yx, px[2] * scale_yx], dtype=float)
    if limits is not None:
        um = dt_limit_shifts_zyx(um, limits)
    if dampening is not None:
        um = um * np.array(dampening, dtype=float)
    return float(um[2]), float(um[1]), float(um[0])
