"""OME-Zarr in/out for the reconstruction CLI (see ``omezarr.py``)."""

from .omezarr import open_ome_zarr  # noqa: F401
