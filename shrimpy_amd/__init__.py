"""shrimpy_amd -- MI355X-native light-sheet reconstruction hot path.

Deskew -> affine registration apply -> Richardson-Lucy deconvolution, as hand-written HIP
kernels for gfx950 behind a C ABI (``include/lsrecon.h``), with a Python host that mirrors the
interface the reference (czbiohub-sf/shrimPy) calls: ``biahub.deskew`` / ``biahub.settings``.

Submodules are imported lazily so that ``import shrimpy_amd`` stays cheap (no torch import).
"""

__version__ = "0.1.0"

__all__ = ["deskew", "register", "deconvolve", "settings", "geometry", "preprocessing", "pipeline"]
