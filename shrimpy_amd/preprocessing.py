"""Adapter that lets a caller written against the reference's preprocessor use the HIP hot path.

The reference builds its per-volume preprocessor with ``build_preprocessor(...)`` and then calls the
returned object with a raw ``(Z, Y, X)`` stack (``shrimpy/preprocessing.py:85-158, 284-366``).
INTEGRATION.md binds this package underneath THAT code (as ``biahub.deskew``), so the reference's
own module stays the caller; this file is for users who want the same call shape without the
reference installed.  It is deliberately small:

* which stages run is decided once, at construction (``flatfield`` and/or ``deskew``; ``phase`` and
  ``vs`` are outside SURVEY.md section 8 and are refused up front);
* the stack is uploaded once -- camera ``uint16`` stays ``uint16`` over PCIe and in HBM, the kernels
  widen it exactly;
* when both stages are on, the flat-field median is its own launch and the division happens while
  the deskew kernel stages its input slab (``csrc/deskew.hip``), so the corrected raw volume is never
  written;
* the device is resolved the reference's way (``cuda`` when a HIP device is visible, else ``cpu``,
  ``shrimpy/preprocessing.py:78-82``): on ``cpu`` every stage runs its native host twin
  (``csrc/host_twins.hip``, bit-equal to the kernels) -- unless ``require_gpu`` is set, which then raises in
  ``warm_up`` as the reference's check does (``:357-363``).

Contract kept from the reference (names are its API): ``build_preprocessor``'s signature, the
``RECON_STEPS`` tuple (``:41``), the returned dict keyed by ``output_channel`` with the optional
``"deskew"`` intermediate (``:347-355``), ``warm_up`` replacing the working shape by the deskewed
one (``:224-244``), and errors propagating to the caller after being logged (``:376-383``).
"""

from __future__ import annotations

import inspect
import logging
import time

import numpy as np

log = logging.getLogger(__name__)

RECON_STEPS = ("flatfield", "deskew", "phase", "vs")   # the reference's step vocabulary (``:41``)
_HOT_PATH_STEPS = frozenset({"flatfield", "deskew"})


def accepted_kwargs(func, settings) -> dict:
    """``settings.model_dump()`` cut down to the parameters ``func`` declares.

    The reference hands a settings model to low-level callables this way (``:44-56``), which makes
    the callee's parameter names the interface; ours are checked against that in
    ``tests/golden/ref_preprocessing.npz``.
    """
    names = inspect.signature(func).parameters.keys()
    dumped = settings.model_dump()
    return {name: dumped[name] for name in dumped if name in names}


_settings_kwargs = accepted_kwargs   # the reference's name for the same helper


def _resolve_device():
    """``cuda`` (HIP under ROCm) when a device is visible, else ``cpu`` -- the reference's rule
    (``shrimpy/preprocessing.py:78-82``); on ``cpu`` the deskew runs its native host twin."""
    import torch

    return torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")


def build_preprocessor(zyx_shape, preprocessing, deskew=None, phase=None, virtual_staining=None,
                       output_channel="phase", require_gpu=False):
    """Reference-shaped factory: ``None`` when ``preprocessing`` names no reconstruction step,
    otherwise a warmed-up :class:`HotPathPreprocessor`.

    ``deskew`` is the plain dict the reference passes (pixel size and scan step already injected by
    its caller, ``shrimpy/dynatrack/manager.py:297-299``); ``phase`` / ``virtual_staining`` are
    accepted for signature compatibility and must stay unused.
    """
    wanted = [s for s in (preprocessing or ()) if s in RECON_STEPS]
    if not wanted:
        return None
    foreign = [s for s in wanted if s not in _HOT_PATH_STEPS]
    if foreign:
        raise NotImplementedError(
            f"{foreign}: phase reconstruction (waveorder) and virtual staining (cytoland) are not part "
            "of the MI355X hot path; keep those steps on the reference's own preprocessor")
    settings = None
    if "deskew" in wanted and deskew:
        from .settings import DeskewSettings

        settings = DeskewSettings(**deskew)
    pre = HotPathPreprocessor(zyx_shape, deskew_settings=settings, output_channel=output_channel,
                              apply_flatfield="flatfield" in wanted, require_gpu=require_gpu)
    pre.warm_up()
    return pre


class HotPathPreprocessor:
    """``pre(stack) -> {output_channel: device tensor}`` for flat-field and deskew on one HIP device."""

    def __init__(self, zyx_shape, deskew_settings=None, output_channel="phase", apply_flatfield=False,
                 require_gpu=False, phase_settings=None, vs_config=None):
        if phase_settings is not None or vs_config is not None:
            raise NotImplementedError("phase / virtual staining settings given to the hot-path adapter")
        self._zyx_shape = tuple(int(n) for n in zyx_shape)
        self._deskew_settings = deskew_settings
        self._apply_flatfield = bool(apply_flatfield)
        self._output_channel = output_channel
        self._require_gpu = bool(require_gpu)
        self._device = None
        self._pending_flat_field = None

    # ------------------------------------------------------------------ set-up
    def warm_up(self) -> None:
        """Pick the device (it must be a GPU as soon as a kernel stage is on) and switch the working
        shape to the deskewed one, which is what downstream consumers size themselves from."""
        self._device = _resolve_device()
        if self._device.type != "cuda" and self._require_gpu:
            raise RuntimeError("no HIP device visible (device resolved to %s) and require_gpu is set; without it the "
                               "stages run their native host twins" % self._device)
        if self._deskew_settings is not None:
            from .deskew import get_deskewed_data_shape

            raw_shape = self._zyx_shape
            self._zyx_shape, voxel = get_deskewed_data_shape(
                raw_data_shape=raw_shape, **accepted_kwargs(get_deskewed_data_shape, self._deskew_settings))
            log.info("deskew geometry: raw %s -> %s, voxel %s um (ratio %s)", raw_shape, self._zyx_shape,
                     tuple(round(v, 4) for v in voxel),
                     getattr(self._deskew_settings, "px_to_scan_ratio", None))

    # ------------------------------------------------------------------ per volume
    def __call__(self, volume_bf, label="", return_intermediates=False) -> dict:
        tag = f"[{label}] " if label else ""
        volume = self._upload(volume_bf)
        fuse = self._apply_flatfield and self._deskew_settings is not None
        self._pending_flat_field = None
        try:
            if fuse:
                self._pending_flat_field = self._step(tag, "flatfield", self._flat_field_pattern, volume)
            elif self._apply_flatfield:
                volume = self._step(tag, "flatfield", self._flat_field_BF, volume)
            deskewed = None
            if self._deskew_settings is not None:
                volume = deskewed = self._step(tag, "deskew", self._deskew, volume)
        finally:
            self._pending_flat_field = None
        out = {self._output_channel: volume}
        if return_intermediates and deskewed is not None and "deskew" not in out:
            out["deskew"] = deskewed
        if self._require_gpu:
            on_host = sorted(k for k, v in out.items() if v.device.type != "cuda")
            if on_host:
                raise RuntimeError(f"{tag}require_gpu: channels {on_host} ended up in host memory")
        return out

    def _upload(self, stack):
        """One host->device copy.  ``uint16`` camera counts go up unconverted when a kernel stage will
        read them (half the PCIe bytes; the kernels convert exactly); anything else as float32."""
        import torch

        kernels_follow = self._apply_flatfield or self._deskew_settings is not None
        if not isinstance(stack, torch.Tensor):
            stack = np.asarray(stack)
            if stack.dtype == np.uint16 and kernels_follow:
                return torch.as_tensor(stack, device=self._device)
        return torch.as_tensor(stack, device=self._device, dtype=torch.float32)

    @staticmethod
    def _step(tag, name, fn, arg):
        """Run one stage with wall-clock logging; a failure is logged once and re-raised unchanged,
        which is what lets the reference's worker turn it into an error message and carry on."""
        started = time.monotonic()
        try:
            out = fn(arg)
        except Exception as err:
            log.error("%s%s FAILED: %s", tag, name, err)
            raise
        log.info("%s%s done in %.2f s", tag, name, time.monotonic() - started)
        return out

    # ------------------------------------------------------------------ stages
    def _flat_field_BF(self, volume):
        """Whole bright-field correction (median over Z per pixel, divide, restore the mean level:
        the arithmetic of the reference's ``:385-404``) as the two HIP launches of ``flatfield.py``."""
        from .flatfield import flat_field_bf

        return flat_field_bf(volume)

    def _flat_field_pattern(self, volume):
        """Only the median / mean launch; ``_deskew`` picks the pattern up and divides in-kernel."""
        from .flatfield import flat_field_pattern

        return flat_field_pattern(volume)

    def _deskew(self, volume):
        from .deskew import deskew_with_matrix, fast_deskew_zyx
        from .geometry import deskew_geometry

        kw = accepted_kwargs(fast_deskew_zyx, self._deskew_settings)
        pattern = self._pending_flat_field
        if pattern is None:
            return fast_deskew_zyx(raw_data=volume, **kw)
        # The fused form needs the explicit matrix entry point; post-processing switches
        # (orientation) are applied exactly as fast_deskew_zyx would.
        from .deskew import orient_volume

        avg = kw.get("average_n_slices", 1)
        geo = deskew_geometry(tuple(volume.shape), kw["ls_angle_deg"], kw["px_to_scan_ratio"],
                              kw["keep_overhang"], avg)
        out = deskew_with_matrix(volume, geo.matrix_3x4, geo.pre_average_shape, avg, flat_field=pattern,
                                 border=kw.get("border", "constant"))
        return orient_volume(out, kw.get("orientation", "identity"))


_LabelfreePreprocessor = HotPathPreprocessor   # the reference's class name, for callers that use it
