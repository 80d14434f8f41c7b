"""Host-side mirror of the reference's per-volume preprocessor for the steps on the hot path.

Mirrors ``shrimpy/preprocessing.py`` of the reference (same names, argument meaning, step order,
logging and error behaviour) for the steps this package implements:

* ``build_preprocessor``            reference ``:85-158``
* ``_LabelfreePreprocessor.warm_up``  reference ``:209-252`` (deskew part ``:224-244``)
* ``_LabelfreePreprocessor.__call__`` reference ``:284-366``
* ``_step`` / ``_flat_field_BF`` / ``_deskew``  reference ``:368-383`` / ``:385-404`` / ``:406-417``

``deskew`` runs the HIP kernel (``shrimpy_amd.deskew.fast_deskew_zyx``); ``flatfield`` runs the
radix-select median kernel (``shrimpy_amd.flatfield``) and, when a deskew follows, its division is
fused into the deskew kernel (same bits, one 8.6 GB round trip less at config 2).  ``phase`` and ``vs`` (waveorder inverse filter,
cytoland U-Net) are out of this package's scope: asking for them raises
``NotImplementedError`` at build time instead of silently skipping.

Unlike the reference, there is no CPU fallback: without a HIP device ``warm_up`` raises, whatever
``require_gpu`` says (the reference's ``cpu`` branch, ``:80``, would hand CPU tensors to a kernel
that only exists on the GPU).
"""

from __future__ import annotations

import logging
import time as _time

from typing import TYPE_CHECKING

import numpy as np

if TYPE_CHECKING:
    from collections.abc import Callable
    from typing import Any

    import torch

logger = logging.getLogger(__name__)

# Same tuple as the reference (``shrimpy/preprocessing.py:41``).
RECON_STEPS = ("flatfield", "deskew", "phase", "vs")
_UNSUPPORTED_STEPS = ("phase", "vs")


def _settings_kwargs(func: Callable, settings: Any) -> dict[str, Any]:
    """Fields of a pydantic *settings* model that *func* accepts (reference ``:44-56``)."""
    import inspect

    accepted = set(inspect.signature(func).parameters)
    return {k: v for k, v in settings.model_dump().items() if k in accepted}


def _resolve_device() -> torch.device:
    """CUDA (= HIP on ROCm) or CPU, never MPS -- the reference's torch-only branch (``:75-82``)."""
    import torch

    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    logger.info("Preprocessing compute device: %s", device)
    return device


def build_preprocessor(
    zyx_shape: tuple[int, int, int],
    preprocessing: list[str] | None,
    deskew: dict | None = None,
    phase: dict | None = None,
    virtual_staining: dict | None = None,
    output_channel: str = "phase",
    require_gpu: bool = False,
):
    """Build ``(np.ndarray ZYX) -> dict[str, torch.Tensor]`` or ``None`` (no reconstruction step).

    Same contract as the reference: the pixel size / scan step must already be injected into
    ``deskew`` by the caller (``shrimpy/dynatrack/manager.py:297-299``).
    """
    pipeline = preprocessing or []
    if not any(step in pipeline for step in RECON_STEPS):
        return None
    unsupported = [s for s in pipeline if s in _UNSUPPORTED_STEPS]
    if unsupported:
        raise NotImplementedError(
            f"preprocessing steps {unsupported} are outside the MI355X hot path "
            "(phase = waveorder inverse filter, vs = cytoland U-Net); run them with the reference"
        )

    deskew_settings = None
    if "deskew" in pipeline and deskew:
        from .settings import DeskewSettings

        deskew_settings = DeskewSettings(**deskew)

    preprocessor = _LabelfreePreprocessor(
        zyx_shape=zyx_shape,
        apply_flatfield="flatfield" in pipeline,
        deskew_settings=deskew_settings,
        output_channel=output_channel,
        require_gpu=require_gpu,
    )
    preprocessor.warm_up()
    return preprocessor


class _LabelfreePreprocessor:
    """Callable ``preprocessor(volume_bf: np.ndarray) -> dict[str, torch.Tensor]``."""

    def __init__(
        self,
        zyx_shape: tuple[int, int, int],
        deskew_settings: Any | None,
        output_channel: str,
        apply_flatfield: bool = False,
        require_gpu: bool = False,
        phase_settings: Any | None = None,
        vs_config: dict[str, Any] | None = None,
    ) -> None:
        if phase_settings is not None or vs_config is not None:
            raise NotImplementedError("phase / virtual staining are outside the MI355X hot path")
        self._zyx_shape = tuple(zyx_shape)
        self._apply_flatfield = apply_flatfield
        self._deskew_settings = deskew_settings
        self._output_channel = output_channel
        self._require_gpu = require_gpu
        self._device = None

    def warm_up(self) -> None:
        """Resolve the device and the deskewed shape before acquisition starts (reference ``:209-244``)."""
        self._device = _resolve_device()
        if self._device.type == "cpu" and (self._require_gpu or self._deskew_settings is not None
                                           or self._apply_flatfield):
            raise RuntimeError(
                "GPU required but none detected: the preprocessing compute device resolved to CPU. "
                "The flat-field and deskew kernels run only on a HIP device (MI355X) -- there is no "
                "CPU fallback."
            )
        if self._deskew_settings is not None:
            from .deskew import get_deskewed_data_shape

            deskewed_shape, _ = get_deskewed_data_shape(
                raw_data_shape=self._zyx_shape,
                **_settings_kwargs(get_deskewed_data_shape, self._deskew_settings),
            )
            logger.info(
                "Preprocessing: deskew will reshape %s -> %s "
                "(px_to_scan_ratio=%s, pixel_size_um=%s, scan_step_um=%s)",
                self._zyx_shape,
                deskewed_shape,
                getattr(self._deskew_settings, "px_to_scan_ratio", None),
                getattr(self._deskew_settings, "pixel_size_um", None),
                getattr(self._deskew_settings, "scan_step_um", None),
            )
            self._zyx_shape = deskewed_shape

    def __call__(
        self,
        volume_bf: np.ndarray,
        label: str = "",
        return_intermediates: bool = False,
    ) -> dict[str, torch.Tensor]:
        """Raw ``(Z, Y, X)`` stack -> ``{output_channel: tensor}`` (+ ``'deskew'`` intermediate)."""
        import torch

        pfx = f"[{label}] " if label else ""
        channels: dict[str, torch.Tensor] = {}

        # one host->device copy; every step then stays on the device (reference :316).  A uint16
        # camera stack is uploaded as it is -- half the PCIe bytes of the reference's float32 copy --
        # and converted (exactly) inside the flat-field / deskew kernels.
        arr = np.asarray(volume_bf) if not isinstance(volume_bf, torch.Tensor) else None
        if arr is not None and arr.dtype == np.uint16 and (self._deskew_settings is not None
                                                           or self._apply_flatfield):
            volume = torch.as_tensor(arr, device=self._device)
        else:
            volume = torch.as_tensor(volume_bf, device=self._device, dtype=torch.float32)

        self._pending_flat_field = None
        if self._apply_flatfield and self._deskew_settings is not None:
            # the median now, the division inside the deskew kernel (bit-identical, see flatfield.py)
            self._pending_flat_field = self._step(pfx, "flatfield", self._flat_field_pattern, volume)
        elif self._apply_flatfield:
            volume = self._step(pfx, "flatfield", self._flat_field_BF, volume)

        volume_deskewed = None
        if self._deskew_settings is not None:
            volume = self._step(pfx, "deskew", self._deskew, volume)
            volume_deskewed = volume
            self._pending_flat_field = None

        channels[self._output_channel] = volume
        if return_intermediates and volume_deskewed is not None:
            channels.setdefault("deskew", volume_deskewed)

        if self._require_gpu:
            offenders = [n for n, t in channels.items() if t.device.type == "cpu"]
            if offenders:
                raise RuntimeError(
                    f"{pfx}GPU required but preprocessing output is on CPU "
                    f"(channels {offenders}); a reconstruction step fell back to CPU."
                )
        return channels

    @staticmethod
    def _step(pfx: str, name: str, fn, arg):
        """Run one step; log ``<label> <name> ok (<t>s)`` or ``FAILED: <error>`` and re-raise."""
        t0 = _time.monotonic()
        try:
            result = fn(arg)
        except Exception as exc:
            logger.error("%s%s FAILED: %s", pfx, name, exc)
            raise
        logger.info("%s%s ok (%.1fs)", pfx, name, _time.monotonic() - t0)
        return result

    def _flat_field_BF(self, volume: torch.Tensor) -> torch.Tensor:
        """Bright-field flat-field: divide out the per-pixel median over Z, keep its mean.

        The reference's torch expression (``:403-404``: ``quantile(0.5, dim=0)``, then
        ``volume / pattern * pattern.mean()``) as HIP kernels (``shrimpy_amd.flatfield``); the
        tensor must live on the GPU -- no CPU fallback.
        """
        from .flatfield import flat_field_bf

        return flat_field_bf(volume)

    def _flat_field_pattern(self, volume: torch.Tensor):
        """The median / mean half of ``_flat_field_BF`` (the division rides along with the deskew)."""
        from .flatfield import flat_field_pattern

        return flat_field_pattern(volume)

    def _deskew(self, volume: torch.Tensor) -> torch.Tensor:
        """``fast_deskew_zyx`` on the device, kwargs filtered by signature (reference ``:406-417``)."""
        from .deskew import fast_deskew_zyx

        logger.debug("Preprocessing: deskewing volume %s...", tuple(volume.shape))
        kwargs = _settings_kwargs(fast_deskew_zyx, self._deskew_settings)
        pending = getattr(self, "_pending_flat_field", None)
        if pending is not None:
            # same geometry as fast_deskew_zyx, with the flat-field division fused into the kernel
            from .deskew import deskew_with_matrix
            from .geometry import deskew_geometry

            geo = deskew_geometry(tuple(volume.shape), kwargs["ls_angle_deg"], kwargs["px_to_scan_ratio"],
                                  kwargs["keep_overhang"], kwargs.get("average_n_slices", 1))
            result = deskew_with_matrix(volume, geo.matrix_3x4, geo.pre_average_shape,
                                        kwargs.get("average_n_slices", 1), flat_field=pending)
        else:
            result = fast_deskew_zyx(raw_data=volume, **kwargs)
        logger.debug("Preprocessing: deskew %s -> %s", tuple(volume.shape), tuple(result.shape))
        return result
