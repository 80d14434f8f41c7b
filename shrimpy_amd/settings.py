"""Settings models of the reconstruction path (pydantic, ``extra="forbid"``, YAML-loadable).

``DeskewSettings`` mirrors ``biahub.settings.DeskewSettings`` as the reference uses it:
constructed from a plain dict (``shrimpy/preprocessing.py:137-140``), dumped with
``.model_dump()`` and filtered to the callee's signature (``:44-56``), attributes
``px_to_scan_ratio`` / ``pixel_size_um`` / ``scan_step_um`` read by ``getattr`` (``:240-242``).
Field names are evidenced at ``shrimpy/dynatrack/tracking.py:200-204``,
``shrimpy/dynatrack/manager.py:297-299`` (scale injection) and
``config/mda/mantis/dynatrack_demo.yaml:161-164``; the ratio rounding rule at
``scripts/measure_psf.py:225``.  The strict-validation idiom is the reference's own
(``shrimpy/config.py:82-127``).

``RegisterSettings`` and ``DeconvolveSettings`` have no reference counterpart (registration and
deconvolution are "being developed", ``docs/data_structure.md:58-62``); they follow the same idiom.
"""

from __future__ import annotations

from pathlib import Path
from typing import Union, Literal, Optional

import numpy as np
import yaml

from pydantic import (
    BaseModel,
    ConfigDict,
    NonNegativeInt,
    PositiveFloat,
    PositiveInt,
    field_validator,
    model_validator,
)


class _StrictModel(BaseModel):
    model_config = ConfigDict(extra="forbid")

    @classmethod
    def from_yaml(cls, path: str | Path):
        """Load and validate a YAML settings file."""
        with open(path) as f:
            raw = yaml.safe_load(f)
        if not isinstance(raw, dict):
            raise ValueError(f"{path}: expected a mapping at the top level")
        return cls(**raw)

    def to_yaml(self, path: str | Path) -> None:
        with open(path, "w") as f:
            yaml.safe_dump(self.model_dump(mode="json"), f, sort_keys=False)


class DeskewSettings(_StrictModel):
    """Oblique light-sheet deskew parameters.

    Either ``px_to_scan_ratio`` or ``scan_step_um`` must be given; the ratio is derived as
    ``round(pixel_size_um / scan_step_um, 3)`` when absent.
    """

    pixel_size_um: PositiveFloat
    ls_angle_deg: PositiveFloat
    px_to_scan_ratio: Optional[PositiveFloat] = None
    scan_step_um: Optional[PositiveFloat] = None
    keep_overhang: bool = False
    average_n_slices: PositiveInt = 3
    # Switches over the two conventions SURVEY.md section 8 marks [RECALLED]; the defaults are the
    # canonical output.  They reach ``fast_deskew_zyx`` / ``get_deskewed_data_shape`` through the
    # reference's signature filter like every other field.
    orientation: str = "identity"
    border: Literal["constant", "grid-constant"] = "constant"
    # the value outside the stack (scipy's cval): a number, or "min" = the stack's minimum ([RECALLED] what biahub's
    # deskew_data fills with when its cval is None, so None is taken as "min" too)
    cval: Union[float, Literal["min"], None] = 0.0

    @field_validator("orientation")
    @classmethod
    def _check_orientation(cls, v: str) -> str:
        from .geometry import parse_orientation

        parse_orientation(v)
        return v

    @field_validator("ls_angle_deg")
    @classmethod
    def _angle_range(cls, v: float) -> float:
        if v > 45:
            raise ValueError("light-sheet angle must be in (0, 45] degrees")
        return round(float(v), 2)

    @field_validator("px_to_scan_ratio")
    @classmethod
    def _round_ratio(cls, v):
        return None if v is None else round(float(v), 3)

    @model_validator(mode="after")
    def _derive_ratio(self):
        if self.px_to_scan_ratio is None:
            if self.scan_step_um is None:
                raise ValueError(
                    "if px_to_scan_ratio is not provided, both pixel_size_um and scan_step_um "
                    "must be provided"
                )
            ratio = round(self.pixel_size_um / self.scan_step_um, 3)
            if not ratio > 0:
                raise ValueError("derived px_to_scan_ratio rounds to zero")
            object.__setattr__(self, "px_to_scan_ratio", ratio)
        return self


class RegisterSettings(_StrictModel):
    """Affine registration apply (label-free <-> fluorescence).

    ``affine_transform_zyx`` is a homogeneous 4x4 in ZYX voxel units mapping TARGET (output)
    coordinates to SOURCE (moving) coordinates -- the ``scipy.ndimage.affine_transform`` convention.

    ``source_channel_names``: the channels the transform is applied to (empty = every channel);
    any other channel -- ``target_channel_name`` among them -- passes through unwarped
    (``cli._channel_plan``).

    ``keep_overhang`` ([RECALLED] biahub ``register``): ``False`` crops the result to the target grid
    (``output_shape_zyx``, default the moving volume's own shape); ``True`` grows the output to the UNION of the target
    grid and the moving volume's footprint in target coordinates, so nothing of either is lost: the box's lower corner
    is folded into the matrix (``resolved``), the output shape is the box's.
    """

    source_channel_names: list[str] = []
    target_channel_name: Optional[str] = None
    affine_transform_zyx: list[list[float]]
    output_shape_zyx: Optional[tuple[PositiveInt, PositiveInt, PositiveInt]] = None
    mode: Literal["constant", "grid-constant"] = "constant"
    cval: float = 0.0
    keep_overhang: bool = False

    def resolved(self, source_shape_zyx):
        """``(matrix_4x4, output_shape_zyx, origin_zyx)`` this settings object applies to a moving volume of
        ``source_shape_zyx``: the matrix maps OUTPUT indices to source coordinates, ``origin_zyx`` is where output
        index 0 sits in target coordinates (all zero unless ``keep_overhang``)."""
        m = np.asarray(self.affine_transform_zyx, dtype=np.float64)
        src = tuple(int(v) for v in source_shape_zyx)
        tgt = tuple(int(v) for v in (self.output_shape_zyx or src))
        if not self.keep_overhang:
            return m, tgt, (0, 0, 0)
        a, t = m[:3, :3], m[:3, 3]
        if abs(np.linalg.det(a)) < 1e-12:
            raise ValueError("keep_overhang needs an invertible affine_transform_zyx")
        # the moving volume's voxel centres span [0, n - 1] per axis: its corners in target coordinates
        corners = np.array([[z, y, x] for z in (0, src[0] - 1) for y in (0, src[1] - 1) for x in (0, src[2] - 1)], dtype=np.float64)
        in_target = (np.linalg.inv(a) @ (corners - t).T).T
        lo = np.minimum(0.0, np.floor(in_target.min(axis=0) + 1e-9)).astype(np.int64)
        hi = np.maximum(np.asarray(tgt) - 1.0, np.ceil(in_target.max(axis=0) - 1e-9)).astype(np.int64)
        shape = tuple(int(v) for v in (hi - lo + 1))
        if max(shape) >= 1 << 30:
            raise ValueError(f"keep_overhang: the union box {shape} is absurd (a near-singular transform?)")
        grown = m.copy()
        grown[:3, 3] = a @ lo.astype(np.float64) + t        # output index i is target coordinate i + lo
        return grown, shape, tuple(int(v) for v in lo)

    @field_validator("affine_transform_zyx")
    @classmethod
    def _check_affine(cls, v):
        m = np.asarray(v, dtype=np.float64)
        if m.shape != (4, 4):
            raise ValueError(f"affine_transform_zyx must be 4x4, got {m.shape}")
        if not np.all(np.isfinite(m)):
            raise ValueError("affine_transform_zyx contains non-finite entries")
        if not np.allclose(m[3], [0, 0, 0, 1]):
            raise ValueError("last row of affine_transform_zyx must be [0, 0, 0, 1]")
        return m.tolist()


class DeconvolveSettings(_StrictModel):
    """Richardson-Lucy deconvolution.

    The PSF comes from ``psf_path`` -- a ``.npy`` ZYX array, or an OME-Zarr store (an averaged bead
    volume as the PSF-characterisation tools around the reference write them with iohub,
    ``scripts/measure_psf.py:273-287``: HCS layout, first position, array ``"0"``, T = C = 0) -- or, when
    absent, is the separable anisotropic Gaussian ``gaussian_sigma_zyx`` truncated to
    ``gaussian_shape_zyx``.  ``psf_shape_zyx`` (odd) cuts a measured PSF around its brightest voxel and renormalises
    it to sum 1 (``load_psf``).  PSFs of up to 15 taps per axis run through the stencil kernels; larger dense ones -- the
    15 x 18 x 18 ... 30 x 36 x 18 bead patches of ``scripts/measure_psf.py:187-190`` -- run the iteration in the
    Fourier domain (``method``, ``shrimpy_amd/deconvolve_fft.py``), at a cost that does not depend on their size.

    ``separable="auto"`` picks the cheapest exact form of the PSF: three 1-D kernels (one fused launch
    per iteration), else ``ky (x) kzx`` -- a y kernel times a dense (z, x) stencil, the shape of a
    tilted light-sheet PSF -- else the dense stencil.  A factorisation is accepted when it reproduces
    the PSF to ``separable_rtol`` of its peak: the default only admits PSFs that factor exactly; a
    measured PSF can be run in a factored form by raising it (the PSF used is then the factored
    one).  ``"force"`` insists on three 1-D kernels, ``"never"`` on the dense stencil.
    """

    iterations: NonNegativeInt = 20
    eps: PositiveFloat = 1e-6
    psf_path: Optional[str] = None
    psf_shape_zyx: Optional[tuple[PositiveInt, PositiveInt, PositiveInt]] = None
    gaussian_sigma_zyx: tuple[PositiveFloat, PositiveFloat, PositiveFloat] = (2.0, 1.2, 1.2)
    gaussian_shape_zyx: tuple[PositiveInt, PositiveInt, PositiveInt] = (9, 7, 7)
    separable: Literal["auto", "force", "never"] = "auto"
    separable_rtol: PositiveFloat = 1e-6
    # "auto": stencil kernels where a tuned one takes the PSF, the Fourier-domain iteration for dense PSFs beyond
    # them (deconvolve.make_plan); "direct" / "fft" insist on one
    method: Literal["auto", "direct", "fft"] = "auto"

    @field_validator("gaussian_shape_zyx")
    @classmethod
    def _odd_taps(cls, v):
        if v is not None and (any(n % 2 == 0 for n in v) or v[0] > 31 or max(v[1:]) > 15):
            raise ValueError("the Gaussian's extents must be odd, <= 15 in plane and <= 31 along z")
        return v

    @field_validator("psf_shape_zyx")
    @classmethod
    def _odd_cut(cls, v):
        if v is not None and (any(n % 2 == 0 for n in v) or max(v) > 129):
            raise ValueError("psf_shape_zyx must be odd and <= 129 per axis (beyond 15 taps per axis -- 31 along z for a "
                             "PSF that factors -- the iteration runs in the Fourier domain)")
        return v

    def load_psf(self):
        """The PSF array this block names (``None`` for the Gaussian): float32 ZYX, cut to ``psf_shape_zyx``
        around its peak when that is given (then non-negative and summing to 1)."""
        import numpy as np

        if not self.psf_path:
            return None
        path = str(self.psf_path)
        if path.endswith(".npy"):
            psf = np.load(path)
        else:
            from pathlib import Path

            from .io.omezarr import as_volume_array, open_ome_zarr

            if not Path(path).is_dir():
                raise FileNotFoundError(f"psf_path {path}: neither a .npy file nor an OME-Zarr store")
            with open_ome_zarr(path, prefer_iohub=False) as store:
                _, pos = next(iter(store.positions()))
                psf = as_volume_array(pos["0"]).read_volume(0, 0)
        psf = np.asarray(psf, dtype=np.float32)
        while psf.ndim > 3 and psf.shape[0] == 1:
            psf = psf[0]
        if psf.ndim != 3:
            raise ValueError(f"psf_path {path}: expected a (Z, Y, X) volume, got shape {psf.shape}")
        if self.psf_shape_zyx is not None:
            want = tuple(int(n) for n in self.psf_shape_zyx)
            peak = np.unravel_index(int(np.argmax(psf)), psf.shape)
            lo = [c - n // 2 for c, n in zip(peak, want)]
            if any(a < 0 or a + n > s for a, n, s in zip(lo, want, psf.shape)):
                raise ValueError(f"psf_path {path}: a {want} window around the peak at {tuple(int(c) for c in peak)} "
                                 f"does not fit the {psf.shape} volume")
            psf = np.clip(psf[tuple(slice(a, a + n) for a, n in zip(lo, want))], 0.0, None)
            total = float(psf.sum(dtype=np.float64))
            if not total > 0:
                raise ValueError(f"psf_path {path}: the cut PSF is empty")
            psf = (psf / total).astype(np.float32)
        elif max(psf.shape) > 129 or (self.method == "direct" and max(psf.shape) > 15):
            raise ValueError(f"psf_path {path}: shape {psf.shape} exceeds "
                             + ("the stencil kernels' 15 taps per axis (method='direct')" if max(psf.shape) <= 129
                                else "129 taps per axis")
                             + "; give psf_shape_zyx (odd) to cut it around its peak")
        return np.ascontiguousarray(psf, dtype=np.float32)


class ReconstructSettings(_StrictModel):
    """Whole per-volume pipeline: (flat-field) -> deskew -> (register) -> (deconvolve)."""

    flatfield: bool = False  # bright-field only: divide out the per-pixel median over Z (reference
    #                          ``RECON_STEPS[0]``, ``shrimpy/preprocessing.py:320-327``)
    deskew: Optional[DeskewSettings] = None
    registration: Optional[RegisterSettings] = None
    deconvolution: Optional[DeconvolveSettings] = None
