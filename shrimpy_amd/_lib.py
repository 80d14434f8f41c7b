"""ctypes binding of ``liblsrecon.so`` (C ABI: ``include/lsrecon.h``).

The extension is the product: nothing here routes through a reference implementation or the test oracle.
A tensor on a HIP device runs the kernels; a CPU tensor (the reference's own ``cpu`` branch,
``shrimpy/preprocessing.py:78-82``) runs the native host twins of the same entry points (``lsr_*_cpu``,
``csrc/host_twins.hip``: the same arithmetic, bit-equal results).  If the shared library is missing, stale
(its source stamp differs from this checkout's) or cannot be loaded, or a device-only object is handed a CPU
tensor, the call fails loudly with :class:`LsrError` -- there is no silent fallback of any kind.

PyTorch is plumbing only: tensors own the device memory (so callers can ``clone()`` them and
``torch.cuda.empty_cache()`` as the reference does, ``shrimpy/dynatrack/tracking.py:1097-1102``)
and ``torch.cuda.current_stream()`` provides the ``hipStream_t`` every entry point takes.
"""

from __future__ import annotations

import ctypes
import os
import subprocess
import threading

from pathlib import Path

CSRC_DIR = Path(__file__).resolve().parent / "csrc"
# LSR_LIBRARY: another build of the same library (a sanitizer build of the host code, a probe build) -- measurement
# and debugging only; the product loads the in-tree one
LIB_PATH = Path(os.environ["LSR_LIBRARY"]) if os.environ.get("LSR_LIBRARY") else CSRC_DIR / "liblsrecon.so"
HEADER_PATH = Path(__file__).resolve().parent.parent / "include" / "lsrecon.h"

# include/lsrecon.h constants
MODE_CONSTANT = 0
MODE_GRID_CONSTANT = 1
MODE_F32_INTERP = 256
EPI_NONE = 0
EPI_RATIO = 1
EPI_UPDATE = 2
EPI_SCALE = 3
E_UNSUPPORTED = -3

_c_f32p = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int
_f32 = ctypes.c_float
_stream = ctypes.c_void_p
_f64p = ctypes.POINTER(ctypes.c_double)

# symbol -> argtypes; every function returns int except lsr_last_error.
SIGNATURES: dict[str, list] = {
    "lsr_version": [],
    "lsr_deskew_f32": [_c_f32p, _i64, _i64, _i64, _c_f32p, _i64, _i64, _i64, _i64, _i64, _i64, _f64p, _int, _stream],
    "lsr_deskew_border": [ctypes.c_void_p, _int, _i64, _i64, _i64, _c_f32p, _i64, _i64, _i64, _i64, _i64, _i64, _f64p,
                          _int, _int, ctypes.c_void_p, ctypes.c_void_p, _stream],
    "lsr_deskew_cval": [ctypes.c_void_p, _int, _i64, _i64, _i64, _c_f32p, _i64, _i64, _i64, _i64, _i64, _i64, _f64p,
                        _int, _int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _stream],
    "lsr_minmax_u16": [ctypes.c_void_p, _i64, _c_f32p, ctypes.c_void_p, _stream],
    "lsr_deskew_u16": [ctypes.c_void_p, _i64, _i64, _i64, _c_f32p, _i64, _i64, _i64, _i64, _i64, _i64, _f64p, _int, _stream],
    "lsr_deskew_flat_f32": [_c_f32p, _i64, _i64, _i64, _c_f32p, _i64, _i64, _i64, _i64, _i64, _i64, _f64p, _int,
                            _c_f32p, _c_f32p, _stream],
    "lsr_deskew_flat_u16": [ctypes.c_void_p, _i64, _i64, _i64, _c_f32p, _i64, _i64, _i64, _i64, _i64, _i64, _f64p, _int,
                            _c_f32p, _c_f32p, _stream],
    "lsr_flatfield_pattern_u16": [ctypes.c_void_p, _i64, _i64, _i64, _c_f32p, _c_f32p, ctypes.c_void_p, _stream],
    "lsr_flatfield_apply_u16": [ctypes.c_void_p, _c_f32p, _c_f32p, _c_f32p, _i64, _i64, _i64, _stream],
    "lsr_flatfield_scratch_bytes": [],
    "lsr_flatfield_pattern_f32": [_c_f32p, _i64, _i64, _i64, _c_f32p, _c_f32p, ctypes.c_void_p, _stream],
    "lsr_flatfield_apply_f32": [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _i64, _i64, _i64, _stream],
    "lsr_affine_f32": [_c_f32p, _i64, _i64, _i64, _c_f32p, _i64, _i64, _i64, _f64p, _f32, _int, _stream],
    "lsr_reduce_scratch_bytes": [],
    "lsr_minmax_f32": [_c_f32p, _i64, _c_f32p, ctypes.c_void_p, _stream],
    "lsr_histogram_f32": [_c_f32p, _i64, _f32, _f32, _int, ctypes.c_void_p, _stream],
    "lsr_weighted_centroid_f32": [_c_f32p, _i64, _i64, _i64, _f32, ctypes.c_void_p, ctypes.c_void_p, _stream],
    "lsr_mask_centroid_f32": [_c_f32p, _i64, _i64, _i64, _f32, ctypes.c_void_p, ctypes.c_void_p, _stream],
    "lsr_blur_reflect_f32": [_c_f32p, _c_f32p, _i64, _i64, _i64, _int, _c_f32p, _int, _f32, _f32, _stream],
    "lsr_match_shape_f32": [_c_f32p, _i64, _i64, _i64, _c_f32p, _i64, _i64, _i64, _stream],
    "lsr_cross_power_c64": [_c_f32p, _c_f32p, _i64, _stream],
    "lsr_rfft_rows_supported": [_i64],
    "lsr_rfft_rows_scratch_bytes": [_i64, _i64],
    "lsr_rfft_rows_t_c64": [_c_f32p, _i64, _i64, _i64, _c_f32p, _i64, _i64, _i64, _c_f32p, _c_f32p, _stream],
    "lsr_irfft_rows_peak": [_c_f32p, _i64, _i64, _i64, _c_f32p, _c_f32p, ctypes.c_void_p, ctypes.c_void_p, _stream],
    "lsr_rfft_rows_zero_t_c64": [_c_f32p, _i64, _i64, _i64, _c_f32p, _i64, _i64, _i64, _c_f32p, _c_f32p, _stream],
    "lsr_spectrum_multiply_z_c64": [_c_f32p, _c_f32p, _c_f32p, _i64, _i64, _i64, _int, _i64, _i64, _stream],
    "lsr_irfft_rows_rl_f32": [_c_f32p, _i64, _i64, _i64, _c_f32p, _c_f32p, _int, _c_f32p, _c_f32p, _i64, _i64, _i64, _f32,
                              _f32, _int, _int, _int, ctypes.c_void_p, _f32, ctypes.c_void_p, _stream],
    "lsr_rl_rows_chain_f32": [_c_f32p, _i64, _i64, _i64, _c_f32p, _c_f32p, _int, _c_f32p, _c_f32p, _i64, _i64, _i64, _f32,
                              _f32, _int, _int, _int, ctypes.c_void_p, _f32, ctypes.c_void_p, _stream],
    "lsr_cross_correlate_z_supported": [_i64],
    "lsr_cross_correlate_z_c64": [_c_f32p, _c_f32p, _c_f32p, _i64, _i64, _i64, _stream],
    "lsr_transpose_last2_c64": [_c_f32p, _c_f32p, _i64, _i64, _i64, _stream],
    "lsr_cross_power_into_c64": [_c_f32p, _c_f32p, _i64, _stream],
    "lsr_peak_abs_shifted_f32": [_c_f32p, _i64, _i64, _i64, ctypes.c_void_p, ctypes.c_void_p, _stream],
    "lsr_affine_kernel_choice": [_i64, _i64, _f64p, _int],
    "lsr_affine_pitched_f32": [_c_f32p, _i64, _i64, _i64, _i64, _i64, _c_f32p, _i64, _i64, _i64, _i64, _i64, _f64p, _f32,
                               _int, _stream],
    "lsr_affine_path_pitched": [_i64, _i64, _i64, _i64, _i64, _f64p, _int],
    "lsr_affine_path": [_i64, _i64, _i64, _f64p, _int],
    "lsr_affine_box_shape": [_i64, _i64, _i64, _f64p, ctypes.POINTER(ctypes.c_int)],
    "lsr_affine_normal_size": [],
    "lsr_affine_normal_blocks": [],
    "lsr_affine_normal_equations_f32": [_c_f32p, _i64, _i64, _i64, _c_f32p, _i64, _i64, _i64, _f64p, ctypes.c_double,
                                        ctypes.c_double, ctypes.POINTER(ctypes.c_int), _f64p, ctypes.c_double,
                                        ctypes.c_void_p, _stream],
    "lsr_blosc_host_codec": [_int],
    "lsr_blosc_decode_host": [ctypes.c_void_p, _i64, ctypes.c_void_p, _i64, ctypes.POINTER(ctypes.c_int)],
    "lsr_blosc_host_encoder": [],
    "lsr_blosc_encode_bound": [_i64, _int, _i64],
    "lsr_blosc_encode_host": [ctypes.c_void_p, _i64, _int, _int, _int, _i64, ctypes.c_void_p, _i64,
                              ctypes.POINTER(ctypes.c_int64)],
    "lsr_blosc_encode_device_plan": [_i64, _int, _i64, _i64, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
                                     ctypes.POINTER(ctypes.c_int64)],
    "lsr_blosc_encode_device": [ctypes.c_void_p, _i64, _int, _i64, _i64, ctypes.c_void_p, _i64, ctypes.c_void_p, _i64,
                                ctypes.c_void_p, _stream],
    "lsr_blosc_decode_device_plan": [_i64, _i64, _i64, _int, ctypes.POINTER(ctypes.c_int64)],
    "lsr_blosc_decode_device": [ctypes.c_void_p, _i64, ctypes.c_void_p, _i64, _i64, _i64, _int, ctypes.c_void_p, _i64,
                                ctypes.c_void_p, _i64, ctypes.c_void_p, _stream],
    "lsr_zstd_lane_decode_cpu": [ctypes.c_void_p, _i64, ctypes.c_void_p, _i64, ctypes.POINTER(ctypes.c_int64)],
    "lsr_correlate_z_max_taps": [],
    "lsr_correlate_z_f32": [_c_f32p, _i64, _i64, _c_f32p, _i64, _i64, _c_f32p, _i64, _i64, _i64, _i64, _i64, _c_f32p, _int,
                            _int, _f32, _c_f32p, _c_f32p, _c_f32p, ctypes.c_void_p, _stream],
    "lsr_crc32c_host": [ctypes.c_void_p, _i64, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)],
    "lsr_crc32c_host_portable": [ctypes.c_void_p, _i64, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)],
    "lsr_average_slices_f32": [_c_f32p, _i64, _i64, _i64, _c_f32p, _i64, _int, _stream],
    "lsr_correlate_sep_f32": [
        _c_f32p, _c_f32p, _c_f32p, _i64, _i64, _i64, _c_f32p, _int, _c_f32p, _int, _c_f32p, _int,
        _int, _f32, _c_f32p, _c_f32p, _c_f32p, _stream,
    ],
    "lsr_correlate_dense_f32": [
        _c_f32p, _c_f32p, _c_f32p, _i64, _i64, _i64, _c_f32p, _int, _int, _int, _int, _f32,
        ctypes.c_void_p, _stream,
    ],
    "lsr_sep_padded_shape": [_i64, _i64, _int, _int, _int, ctypes.POINTER(ctypes.c_int64)],
    "lsr_correlate_sep_strided_f32": [
        _c_f32p, _i64, _i64, _c_f32p, _i64, _i64, _c_f32p, _i64, _i64, _i64, _i64, _i64,
        _c_f32p, _int, _c_f32p, _int, _c_f32p, _int, _int, _f32, _c_f32p, _c_f32p, _c_f32p, _stream,
    ],
    "lsr_rl_sep_f32": [
        _c_f32p, _i64, _i64, _int, _c_f32p, _c_f32p, _c_f32p, _i64, _i64, _i64, _c_f32p, _c_f32p, _int, _c_f32p,
        _c_f32p, _int, _c_f32p, _c_f32p, _int, _c_f32p, _c_f32p, _c_f32p, _int, _f32, _stream,
    ],
    "lsr_rl_sep_fused_supported": [_int, _int, _int],
    "lsr_rl_sep_fused_taps_count": [],
    "lsr_rl_sep_fused_prepare_taps": [ctypes.c_void_p, _int, ctypes.c_void_p, _int, ctypes.c_void_p, _int,
                                      ctypes.c_void_p],
    "lsr_rl_sep_fused_f32": [
        _c_f32p, _i64, _i64, _int, _c_f32p, _c_f32p, _c_f32p, _i64, _i64, _i64, _c_f32p, _int, _int, _int,
        _c_f32p, _c_f32p, _c_f32p, _int, _f32, _stream,
    ],
    "lsr_rl_ysep_fused_supported": [_int, _int, _int],
    "lsr_rl_ysep_fused_taps_count": [],
    "lsr_rl_ysep_fused_prepare_taps": [ctypes.c_void_p, _int, ctypes.c_void_p, _int, _int, ctypes.c_void_p],
    "lsr_rl_ysep_fused_f32": [
        _c_f32p, _i64, _i64, _int, _c_f32p, _c_f32p, _c_f32p, _i64, _i64, _i64, _c_f32p, _int, _int, _int,
        ctypes.c_void_p, _f32, _int, _f32, _stream,
    ],
    "lsr_dense_taps_count": [_int, _int, _int],
    "lsr_dense_prepare_taps": [ctypes.c_void_p, _int, _int, _int, _int, ctypes.c_void_p],
    "lsr_correlate_zxy_padded_f32": [
        _c_f32p, _i64, _i64, _c_f32p, _i64, _i64, _c_f32p, _i64, _i64, _i64, _i64, _i64,
        _c_f32p, _c_f32p, _int, _int, _int, _int, _f32, ctypes.c_void_p, _f32, _stream,
    ],
    "lsr_correlate_dense_padded_f32": [
        _c_f32p, _i64, _i64, _c_f32p, _i64, _i64, _c_f32p, _i64, _i64, _i64, _i64, _i64,
        _c_f32p, _int, _int, _int, _int, _f32, ctypes.c_void_p, _f32, _stream,
    ],
    "lsr_rl_dense_padded_f32": [
        _c_f32p, _i64, _i64, _int, _c_f32p, _c_f32p, _c_f32p, _i64, _i64, _i64, _c_f32p, _c_f32p,
        _int, _int, _int, ctypes.c_void_p, _f32, _int, _f32, _stream,
    ],
    "lsr_rl_dense_f32": [
        _c_f32p, _c_f32p, _c_f32p, _i64, _i64, _i64, _c_f32p, _c_f32p, _int, _int, _int,
        ctypes.c_void_p, _int, _f32, _stream,
    ],
    "lsr_pinned_alloc": [_i64, ctypes.POINTER(ctypes.c_void_p)],
    "lsr_pinned_free": [ctypes.c_void_p],
    "lsr_set_host_threads": [_int],
    "lsr_get_host_threads": [],
}
# the `_stats` forms of the Richardson-Lucy entries: one more argument (double* stats) in front of the stream
for _name in ("lsr_rl_sep_fused_f32", "lsr_rl_sep_f32", "lsr_rl_ysep_fused_f32", "lsr_rl_dense_padded_f32", "lsr_rl_dense_f32",
              "lsr_correlate_sep_f32", "lsr_correlate_dense_f32", "lsr_correlate_sep_strided_f32",
              "lsr_correlate_dense_padded_f32", "lsr_correlate_zxy_padded_f32"):
    SIGNATURES[_name.replace("_f32", "_stats_f32")] = SIGNATURES[_name][:-1] + [ctypes.c_void_p, _stream]
# host twins (csrc/host_twins.hip): the device entry point's signature, host pointers
for _name in ("lsr_deskew_f32", "lsr_deskew_u16", "lsr_deskew_cval", "lsr_affine_f32", "lsr_average_slices_f32", "lsr_correlate_sep_f32",
              "lsr_correlate_dense_f32", "lsr_rl_dense_f32", "lsr_correlate_sep_stats_f32", "lsr_correlate_dense_stats_f32",
              "lsr_rl_dense_stats_f32", "lsr_flatfield_pattern_f32", "lsr_flatfield_pattern_u16",
              "lsr_flatfield_apply_f32", "lsr_flatfield_apply_u16",
              # ... and of the DynaTrack estimators (csrc/estimators_host.hip)
              "lsr_minmax_f32", "lsr_histogram_f32", "lsr_weighted_centroid_f32", "lsr_mask_centroid_f32",
              "lsr_blur_reflect_f32", "lsr_match_shape_f32", "lsr_cross_power_c64", "lsr_cross_power_into_c64",
              "lsr_peak_abs_shifted_f32",
              # ... and of the device-side chunk codecs (csrc/blosc_encode.hip)
              "lsr_blosc_encode_device", "lsr_blosc_decode_device"):
    SIGNATURES[_name + "_cpu"] = SIGNATURES[_name]


def kernel_source_sha16() -> str:
    """Fingerprint of the device code this checkout builds (every ``csrc/*.hip|*.hpp``, the Makefile
    and the C-ABI header).  Measurements that belong to a particular build of the kernels -- the PMC
    traffic record ``bench.py`` reports -- carry it, so a stale record is recognisable after the
    kernels change (the GPU box has no ``.git`` to ask)."""
    import hashlib

    h = hashlib.sha256()
    # (the order csrc/Makefile's STAMP_FILES uses: header, Makefile, then the sources by the bytes of their names)
    files = [HEADER_PATH, CSRC_DIR / "Makefile"] + sorted(list(CSRC_DIR.glob("*.hip")) + list(CSRC_DIR.glob("*.hpp")),
                                                          key=lambda f: f.name.encode())
    for f in files:
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


class LsrError(RuntimeError):
    """A liblsrecon call failed (argument error or HIP launch error)."""

    def __init__(self, func: str, code: int, message: str):
        self.func = func
        self.code = code
        super().__init__(f"{func} failed ({code}): {message}")


class LsrUnsupported(LsrError):
    """The request is valid but outside what this entry point covers (``LSR_E_UNSUPPORTED``)."""


_lock = threading.Lock()
_lib: ctypes.CDLL | None = None


# what the last ``build()`` of this process did: translation units hipcc compiled, whether the library was linked
LAST_BUILD: dict = {}


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile ``liblsrecon.so`` in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    cmd = ["make", "-C", str(CSRC_DIR), "-j", str(min(8, os.cpu_count() or 1))]
    if force:
        subprocess.run(["make", "-C", str(CSRC_DIR), "clean"], check=True, capture_output=not verbose)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"building liblsrecon.so failed:\n{proc.stdout}\n{proc.stderr}")
    if verbose:
        print(proc.stdout)
    lines = [ln for ln in proc.stdout.splitlines() if "hipcc" in ln]
    LAST_BUILD.update(compiled=sum(" -c " in ln for ln in lines), linked=any(" -shared " in ln for ln in lines),
                      forced=bool(force))
    return LIB_PATH


def load() -> ctypes.CDLL:
    """Load the extension (once).  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not LIB_PATH.exists():
            raise LsrError(
                "load", -1,
                f"{LIB_PATH} is missing; build it with `make -C {CSRC_DIR}` or "
                "`python -c 'import __graft_entry__ as g; g.build()'`. Nothing else implements these entry points "
                "(the host twins for CPU tensors live in the same library).",
            )
        # torch ships its own libamdhip64.so.7; import it first so this library binds to the
        # SAME HIP runtime (same SONAME) and streams / pointers are interchangeable.
        import torch  # noqa: F401

        lib = ctypes.CDLL(str(LIB_PATH), mode=ctypes.RTLD_GLOBAL)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        lib.lsr_last_error.argtypes = []
        lib.lsr_last_error.restype = ctypes.c_char_p
        _check_stamp(lib)
        _lib = lib
    return _lib


def library_source_sha16(lib: ctypes.CDLL | None = None) -> str:
    """The stamp compiled into the loaded binary (``lsr_source_sha16``, csrc/Makefile ``SHA16``)."""
    lib = lib if lib is not None else load()
    fn = lib.lsr_source_sha16
    fn.argtypes = []
    fn.restype = ctypes.c_char_p
    return fn().decode("ascii", "replace")


def _check_stamp(lib: ctypes.CDLL) -> None:
    """Refuse a binary that was not built from the sources beside it.  ``liblsrecon.so`` is git-ignored and reaches the
    GPU box as a file, so nothing else ties it to the checkout: an edit without a rebuild, or a probe build left in
    place, would otherwise be measured and tested under the committed tree's name.  ``LSR_ALLOW_STALE_LIBRARY=1``
    turns the refusal into a warning (bisecting an old binary against new host code)."""
    try:
        have = library_source_sha16(lib)
    except AttributeError:
        have = "unstamped"
    want = kernel_source_sha16()
    if have == want:
        return
    msg = (f"{LIB_PATH} was built from sources {have}, this checkout's are {want} (csrc/*.hip, *.hpp, Makefile, "
           f"include/lsrecon.h): rebuild it with `make -C {CSRC_DIR}` or "
           "`python -c 'import __graft_entry__ as g; g.build()'`")
    if os.environ.get("LSR_ALLOW_STALE_LIBRARY") == "1":
        import warnings

        warnings.warn("stale liblsrecon.so in use: " + msg, RuntimeWarning, stacklevel=3)
        return
    raise LsrError("load", -1, msg)


def call(name: str, *args) -> None:
    """Invoke an entry point and raise on a non-zero status."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.lsr_last_error().decode("utf-8", "replace")
        cls = LsrUnsupported if rc == E_UNSUPPORTED else LsrError
        raise cls(name, rc, msg)


def call_value(name: str, *args) -> int:
    """Invoke an entry point whose return value is an answer (a count, a yes/no), not a status."""
    return int(getattr(load(), name)(*args))


def matrix12(matrix_3x4):
    """Row-major 3x4 -> ``double[12]``."""
    import numpy as np

    m = np.ascontiguousarray(np.asarray(matrix_3x4, dtype=np.float64).reshape(12))
    return (ctypes.c_double * 12)(*m.tolist())


def require_device_f32(t, name: str):
    """Validate a tensor handed to the C ABI: HIP device, float32, contiguous."""
    import torch

    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor, got {type(t).__name__}")
    if t.device.type != "cuda":
        raise LsrError(
            "require_device", -1,
            f"{name} is on {t.device}; this entry runs only on a HIP device (MI355X) -- CPU tensors go through "
            "shrimpy_amd.host (the lsr_*_cpu twins) from the public functions, not through the device objects.",
        )
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t


def mark_written(t) -> None:
    """Tell torch that ``t`` was written in place.  The kernels write through raw ``data_ptr()``s, which
    torch's version counter does not see; anything that keys on ``t._version`` (the DynaTrack reference
    cache, autograd's saved-tensor checks) would otherwise take a caller-supplied ``out=`` buffer for
    unchanged.  A zero-length in-place op on a view bumps the shared counter and launches nothing."""
    if t is not None and hasattr(t, "narrow") and t.dim() > 0:
        t.narrow(0, 0, 0).zero_()


def stream_ptr(device=None) -> int:
    """The current torch HIP stream of ``device`` as an integer ``hipStream_t``."""
    import torch

    return int(torch.cuda.current_stream(device).cuda_stream)
