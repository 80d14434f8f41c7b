"""Minimal OME-Zarr (NGFF 0.4 / Zarr v2 and NGFF 0.5 / Zarr v3) reader-writer for 5-D TCZYX data.

The reference reads and writes its data through ``iohub.open_ome_zarr`` (read surface:
``shrimpy/replay_camera.py:176-268`` -- ``layout="auto"``, ``positions()`` yielding
``"row/col/fov"`` keys, array ``"0"``, 5-D TCZYX, ``multiscales[0].datasets[0]
.coordinateTransformations`` scale, index 2 = Z; write surface:
``shrimpy/dynatrack/tracking.py:1337-1367`` -- ``layout="hcs"``, ``create_position``,
``create_zeros(chunks=(1, 1, min(32, nz), ny, nx))``, and ``scripts/measure_psf.py:273-287`` for
the scale transform).  iohub / zarr / numcodecs are not installable in the build image, so this
module implements the subset of that surface the reconstruction CLI needs, with the same names:

    with open_ome_zarr(path, layout="hcs", mode="w", channel_names=[...]) as plate:
        pos = plate.create_position("A", "1", "0")
        arr = pos.create_zeros("0", shape=(T, C, Z, Y, X), dtype="float32", scale=(1, 1, dz, dy, dx))
        arr.write_volume(t, c, zyx)
    with open_ome_zarr(path) as plate:
        for key, pos in plate.positions():
            zyx = pos["0"].read_volume(t, c)

Chunks are stored raw (``bytes`` codec / no compressor) or zlib/gzip-compressed.  Blosc and the
sharding codec (what the acquisition engine writes, ``shrimpy/mantis/mantis_engine.py:474-481``)
need numcodecs/zarr: when ``iohub`` is importable, ``open_ome_zarr`` here simply returns iohub's
object, which handles them.
"""

from __future__ import annotations

import gzip
import itertools
import json
import os
import zlib

from pathlib import Path
from typing import Iterator, Sequence

import numpy as np

__all__ = ["open_ome_zarr", "Plate", "Position", "ZarrArray", "UnsupportedCodec"]

AXES = [
    {"name": "T", "type": "time", "unit": "second"},
    {"name": "C", "type": "channel"},
    {"name": "Z", "type": "space", "unit": "micrometer"},
    {"name": "Y", "type": "space", "unit": "micrometer"},
    {"name": "X", "type": "space", "unit": "micrometer"},
]
_V3_DTYPES = {"float32": "<f4", "float64": "<f8", "uint16": "<u2", "uint8": "|u1", "int16": "<i2",
              "int32": "<i4", "uint32": "<u4"}


class UnsupportedCodec(RuntimeError):
    """The array uses a codec this minimal reader cannot decode (blosc, sharding, ...)."""


def _read_json(p: Path):
    with open(p) as f:
        return json.load(f)


def _write_json(p: Path, obj) -> None:
    p.parent.mkdir(parents=True, exist_ok=True)
    with open(p, "w") as f:
        json.dump(obj, f, indent=1)


class _Node:
    """A Zarr group on disk (v2: .zgroup/.zattrs, v3: zarr.json)."""

    def __init__(self, path: Path, version: str, mode: str):
        self.path, self.version, self.mode = Path(path), version, mode

    # -- attributes -------------------------------------------------------------------------
    @property
    def zattrs(self) -> dict:
        if self.version == "0.5":
            meta = _read_json(self.path / "zarr.json")
            return meta.get("attributes", {}).get("ome", {})
        p = self.path / ".zattrs"
        return _read_json(p) if p.exists() else {}

    def _write_group(self, attrs: dict) -> None:
        if self.mode == "r":
            raise PermissionError("store opened read-only")
        if self.version == "0.5":
            ome = dict(attrs)
            ome.setdefault("version", "0.5")
            _write_json(self.path / "zarr.json",
                        {"zarr_format": 3, "node_type": "group", "attributes": {"ome": ome}})
        else:
            _write_json(self.path / ".zgroup", {"zarr_format": 2})
            _write_json(self.path / ".zattrs", attrs)


class ZarrArray:
    """One N-D array; whole ``(t, c)`` volumes are the unit of I/O."""

    def __init__(self, path: Path, version: str, mode: str):
        self.path, self.version, self.mode = Path(path), version, mode
        if version == "0.5":
            meta = _read_json(self.path / "zarr.json")
            self.shape = tuple(meta["shape"])
            self.chunks = tuple(meta["chunk_grid"]["configuration"]["chunk_shape"])
            self.dtype = np.dtype(_V3_DTYPES.get(meta["data_type"], meta["data_type"]))
            self.fill_value = meta.get("fill_value", 0) or 0
            enc = meta.get("chunk_key_encoding", {"name": "default"})
            self._sep = enc.get("configuration", {}).get("separator", "/")
            self._prefix = "c" + self._sep if enc.get("name", "default") == "default" else ""
            self._compress = None
            for codec in meta.get("codecs", []):
                name = codec["name"] if isinstance(codec, dict) else codec
                if name == "bytes":
                    if codec.get("configuration", {}).get("endian", "little") != "little":
                        raise UnsupportedCodec("big-endian chunks")
                elif name in ("gzip", "zlib"):
                    self._compress = name
                else:
                    raise UnsupportedCodec(f"codec {name!r} in {self.path} needs zarr/numcodecs (iohub)")
        else:
            meta = _read_json(self.path / ".zarray")
            self.shape = tuple(meta["shape"])
            self.chunks = tuple(meta["chunks"])
            self.dtype = np.dtype(meta["dtype"])
            self.fill_value = meta.get("fill_value", 0) or 0
            self._sep = meta.get("dimension_separator", ".")
            self._prefix = ""
            comp = meta.get("compressor")
            if comp is None:
                self._compress = None
            elif comp.get("id") in ("zlib", "gzip"):
                self._compress = comp["id"]
            else:
                raise UnsupportedCodec(f"compressor {comp.get('id')!r} in {self.path} needs numcodecs (iohub)")
            if meta.get("order", "C") != "C" or meta.get("filters"):
                raise UnsupportedCodec("only C-order arrays without filters are supported")

    # -- creation ---------------------------------------------------------------------------
    @classmethod
    def create(cls, path: Path, version: str, shape, chunks, dtype, compress: str | None = None):
        path = Path(path)
        dtype = np.dtype(dtype)
        if version == "0.5":
            codecs = [{"name": "bytes", "configuration": {"endian": "little"}}]
            if compress:
                codecs.append({"name": "gzip", "configuration": {"level": 1}})
            _write_json(path / "zarr.json", {
                "zarr_format": 3, "node_type": "array", "shape": list(shape), "data_type": dtype.name,
                "chunk_grid": {"name": "regular", "configuration": {"chunk_shape": list(chunks)}},
                "chunk_key_encoding": {"name": "default", "configuration": {"separator": "/"}},
                "fill_value": 0, "codecs": codecs,
                "dimension_names": [a["name"] for a in AXES][-len(shape):],
            })
        else:
            _write_json(path / ".zarray", {
                "zarr_format": 2, "shape": list(shape), "chunks": list(chunks), "dtype": dtype.str,
                "compressor": {"id": "zlib", "level": 1} if compress else None, "fill_value": 0,
                "order": "C", "filters": None, "dimension_separator": "/",
            })
        return cls(path, version, "w")

    # -- chunk I/O --------------------------------------------------------------------------
    def _chunk_path(self, idx: Sequence[int]) -> Path:
        return self.path / (self._prefix + self._sep.join(str(i) for i in idx))

    def _read_chunk(self, idx) -> np.ndarray | None:
        p = self._chunk_path(idx)
        if not p.exists():
            return None
        raw = p.read_bytes()
        if self._compress == "gzip":
            raw = gzip.decompress(raw)
        elif self._compress == "zlib":
            raw = zlib.decompress(raw)
        return np.frombuffer(raw, dtype=self.dtype).reshape(self.chunks)

    def _write_chunk(self, idx, block: np.ndarray) -> None:
        raw = np.ascontiguousarray(block, dtype=self.dtype).tobytes()
        if self._compress == "gzip":
            raw = gzip.compress(raw, compresslevel=1)
        elif self._compress == "zlib":
            raw = zlib.compress(raw, 1)
        p = self._chunk_path(idx)
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_bytes(raw)

    def _grid(self, lead: tuple[int, ...]):
        """Chunk indices covering the trailing (Z, Y, X) block at leading indices ``lead``."""
        n = len(self.shape)
        k = len(lead)
        for i, c in zip(lead, self.chunks[:k]):
            if c != 1:
                raise UnsupportedCodec("leading (T, C) chunk sizes other than 1 are not supported")
        ranges = [range(-(-self.shape[d] // self.chunks[d])) for d in range(k, n)]
        return itertools.product(*ranges)

    # Chunks of one volume are independent files: they are read / written by a small thread pool
    # (file I/O and zlib release the GIL), and an uncompressed chunk that is a whole contiguous
    # z-range of the volume -- the layout the acquisition writes, chunks (1, 1, <=32, ny, nx),
    # ``shrimpy/dynatrack/tracking.py:1337-1367`` -- moves between the file and the caller's
    # buffer (e.g. a pinned staging slot) without an intermediate copy.
    _POOL_MIN_BYTES = 8 << 20

    def _map_chunks(self, fn, cidxs, nbytes):
        workers = min(16, os.cpu_count() or 1, len(cidxs))
        if workers <= 1 or nbytes < self._POOL_MIN_BYTES:
            for c in cidxs:
                fn(c)
            return
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(workers, "lsr-zarr") as pool:
            list(pool.map(fn, cidxs))  # list(): re-raise the first worker exception

    def _slab_view(self, vol: np.ndarray, sl) -> memoryview | None:
        """``vol[sl]`` as bytes when that block is a full chunk and contiguous in ``vol``."""
        vchunks = self.chunks[len(self.shape) - 3:]
        full = all(s.stop - s.start == c for s, c in zip(sl, vchunks))
        if not full or self._compress is not None or vol.dtype != self.dtype or not vol.flags.c_contiguous:
            return None
        if (sl[1].start, sl[1].stop, sl[2].start, sl[2].stop) != (0, vol.shape[1], 0, vol.shape[2]):
            return None
        return memoryview(vol[sl]).cast("B")

    def read_volume(self, *lead: int, out: np.ndarray | None = None) -> np.ndarray:
        """The (Z, Y, X) volume at leading indices (t, c) as a C-contiguous array; with ``out``
        (same shape and dtype, e.g. a pinned staging buffer) the chunks are decoded into it."""
        k = len(self.shape) - 3
        if len(lead) != k:
            raise IndexError(f"expected {k} leading indices, got {len(lead)}")
        for i, n in zip(lead, self.shape):
            if not 0 <= i < n:
                raise IndexError(f"index {lead} out of range for shape {self.shape}")
        vshape, vchunks = self.shape[k:], self.chunks[k:]
        if out is None:
            out = np.empty(vshape, dtype=self.dtype)
        elif tuple(out.shape) != tuple(vshape) or out.dtype != self.dtype:
            raise ValueError(f"out must be {tuple(vshape)} {self.dtype}, got {out.shape} {out.dtype}")
        lead = tuple(lead)

        def read_one(cidx):
            sl = tuple(slice(c * s, min((c + 1) * s, n)) for c, s, n in zip(cidx, vchunks, vshape))
            path = self._chunk_path(lead + cidx)
            view = self._slab_view(out, sl)
            if view is not None:
                try:
                    with open(path, "rb", buffering=0) as f:
                        got = 0
                        while got < len(view):
                            n = f.readinto(view[got:])
                            if not n:
                                raise OSError(f"chunk {path} is shorter than {len(view)} bytes")
                            got += n
                except FileNotFoundError:
                    out[sl] = self.fill_value
                return
            block = self._read_chunk(lead + cidx)
            if block is None:
                out[sl] = self.fill_value
                return
            block = block.reshape(vchunks)
            out[sl] = block[tuple(slice(0, s.stop - s.start) for s in sl)]

        self._map_chunks(read_one, list(self._grid(lead)), out.nbytes)
        return out

    def write_volume(self, *args) -> None:
        """``write_volume(t, c, zyx)``: store one (Z, Y, X) volume (edge chunks are zero-padded)."""
        if self.mode == "r":
            raise PermissionError("store opened read-only")
        *lead, vol = args
        k = len(self.shape) - 3
        vol = np.asarray(vol)
        if len(lead) != k or tuple(vol.shape) != tuple(self.shape[k:]):
            raise ValueError(f"expected {k} indices and a volume of shape {self.shape[k:]}, got {vol.shape}")
        vchunks = self.chunks[k:]
        lead = tuple(lead)

        def write_one(cidx):
            sl = tuple(slice(c * s, min((c + 1) * s, n)) for c, s, n in zip(cidx, vchunks, vol.shape))
            view = self._slab_view(vol, sl)
            if view is not None:
                path = self._chunk_path(lead + cidx)
                path.parent.mkdir(parents=True, exist_ok=True)
                with open(path, "wb", buffering=0) as f:
                    done = 0
                    while done < len(view):
                        done += f.write(view[done:])
                return
            block = np.zeros(self.chunks, dtype=self.dtype)
            block.reshape(vchunks)[tuple(slice(0, s.stop - s.start) for s in sl)] = vol[sl]
            self._write_chunk(lead + cidx, block)

        self._map_chunks(write_one, list(self._grid(lead)), vol.nbytes)

    def __getitem__(self, key):
        """Convenience for tests: ``arr[t, c]`` -> volume; ``arr[:]`` -> everything."""
        k = len(self.shape) - 3
        if isinstance(key, tuple) and len(key) == k and all(isinstance(i, (int, np.integer)) for i in key):
            return self.read_volume(*key)
        out = np.empty(self.shape, dtype=self.dtype)
        for lead in itertools.product(*[range(n) for n in self.shape[:k]]):
            out[lead] = self.read_volume(*lead)
        return out[key]


class Position(_Node):
    """One field of view: an NGFF image group with multiscale level arrays ("0", ...)."""

    def __init__(self, path, version, mode, channel_names=None):
        super().__init__(path, version, mode)
        self._channel_names = list(channel_names) if channel_names else None

    @property
    def channel_names(self) -> list[str]:
        if self._channel_names is not None:
            return self._channel_names
        omero = self.zattrs.get("omero", {})
        return [c.get("label", str(i)) for i, c in enumerate(omero.get("channels", []))]

    @property
    def scale(self) -> tuple[float, ...]:
        """(T, C, Z, Y, X) scale of level 0 (ones when absent)."""
        ms = self.zattrs.get("multiscales", [{}])
        ds = ms[0].get("datasets", [{}]) if ms else [{}]
        for t in ds[0].get("coordinateTransformations", []):
            if t.get("type") == "scale":
                return tuple(float(v) for v in t["scale"])
        return (1.0,) * 5

    def __getitem__(self, name: str) -> ZarrArray:
        return ZarrArray(self.path / name, self.version, self.mode)

    @property
    def data(self) -> ZarrArray:
        return self["0"]

    def create_zeros(self, name: str, shape, dtype="float32", chunks=None, scale=None,
                     compress: str | None = None) -> ZarrArray:
        """Create level ``name`` (TCZYX).  Default chunks follow the reference:
        ``(1, 1, min(32, nz), ny, nx)`` (``shrimpy/dynatrack/tracking.py:1362``)."""
        shape = tuple(int(s) for s in shape)
        if len(shape) != 5:
            raise ValueError(f"expected a 5-D TCZYX shape, got {shape}")
        if chunks is None:
            chunks = (1, 1, min(32, shape[2]), shape[3], shape[4])
        arr = ZarrArray.create(self.path / name, self.version, shape, chunks, dtype, compress)
        scale = [float(s) for s in (scale if scale is not None else (1, 1, 1, 1, 1))]
        attrs = {
            "multiscales": [{
                "version": self.version, "axes": AXES, "name": "0",
                "datasets": [{"path": name,
                              "coordinateTransformations": [{"type": "scale", "scale": scale}]}],
            }],
            "omero": {"channels": [{"label": n, "active": True, "color": "FFFFFF",
                                    "window": {"start": 0, "end": 65535, "min": 0, "max": 65535}}
                                   for n in (self._channel_names or [str(i) for i in range(shape[1])])]},
        }
        self._write_group(attrs)
        return arr

    def positions(self) -> Iterator[tuple[str, "Position"]]:
        yield "0/0/0", self

    def close(self) -> None:
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class Plate(_Node):
    """An HCS plate: ``row/col/fov`` positions."""

    def __init__(self, path, version, mode, channel_names=None):
        super().__init__(path, version, mode)
        self._channel_names = list(channel_names) if channel_names else None
        self._wells: dict[str, list[str]] = {}
        if mode == "w":
            self._flush()
        else:
            plate = self.zattrs.get("plate", {})
            for w in plate.get("wells", []):
                wpath = w["path"]
                wnode = _Node(self.path / wpath, version, "r")
                images = wnode.zattrs.get("well", {}).get("images", [])
                self._wells[wpath] = [im["path"] for im in images]

    def _flush(self) -> None:
        rows = sorted({w.split("/")[0] for w in self._wells})
        cols = sorted({w.split("/")[1] for w in self._wells})
        plate = {
            "version": self.version,
            "rows": [{"name": r} for r in rows],
            "columns": [{"name": c} for c in cols],
            "wells": [{"path": w, "rowIndex": rows.index(w.split("/")[0]),
                       "columnIndex": cols.index(w.split("/")[1])} for w in self._wells],
        }
        self._write_group({"plate": plate})
        for w, images in self._wells.items():
            _Node(self.path / w, self.version, "w")._write_group(
                {"well": {"version": self.version, "images": [{"path": i} for i in images]}})
            row = self.path / w.split("/")[0]
            if self.version == "0.5":
                if not (row / "zarr.json").exists():
                    _write_json(row / "zarr.json", {"zarr_format": 3, "node_type": "group", "attributes": {}})
            elif not (row / ".zgroup").exists():
                _write_json(row / ".zgroup", {"zarr_format": 2})

    def create_position(self, row: str, col: str, fov: str) -> Position:
        for part in (row, col, fov):
            if not str(part).isalnum():
                raise ValueError(f"position path components must be alphanumeric, got {part!r}")
        well = f"{row}/{col}"
        self._wells.setdefault(well, [])
        if str(fov) in self._wells[well]:
            raise FileExistsError(f"position {well}/{fov} exists")
        self._wells[well].append(str(fov))
        self._flush()
        return Position(self.path / well / str(fov), self.version, "w", self._channel_names)

    def positions(self) -> Iterator[tuple[str, Position]]:
        for well, images in self._wells.items():
            for fov in images:
                yield f"{well}/{fov}", Position(self.path / well / fov, self.version, self.mode,
                                                self._channel_names)

    def __getitem__(self, key: str) -> Position:
        return dict(self.positions())[key]

    def close(self) -> None:
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def _detect(path: Path) -> tuple[str, str]:
    """(version, layout) of an existing store."""
    if (path / "zarr.json").exists():
        attrs = _read_json(path / "zarr.json").get("attributes", {}).get("ome", {})
        version = "0.5"
    elif (path / ".zgroup").exists() or (path / ".zattrs").exists():
        attrs = _read_json(path / ".zattrs") if (path / ".zattrs").exists() else {}
        version = "0.4"
    else:
        raise FileNotFoundError(f"{path} is not a Zarr group")
    if "plate" in attrs:
        return version, "hcs"
    if "multiscales" in attrs:
        return version, "fov"
    raise ValueError(f"{path}: neither an HCS plate nor an NGFF image")


def open_ome_zarr(store_path, layout: str = "auto", mode: str = "r", channel_names=None,
                  version: str = "0.4", prefer_iohub: bool = True):
    """Open an OME-Zarr store (the subset of ``iohub.open_ome_zarr`` this package uses)."""
    if prefer_iohub:
        try:  # the full implementation (blosc, sharding, dask) when it is installed
            from iohub import open_ome_zarr as _iohub_open

            kw = dict(layout=layout, mode=mode)
            if mode != "r":
                kw.update(channel_names=channel_names, version=version)
            return _iohub_open(str(store_path), **kw)
        except ImportError:
            pass
    path = Path(store_path)
    if mode not in ("r", "w", "a"):
        raise ValueError("mode must be 'r', 'w' or 'a'")
    if mode == "r" or (mode == "a" and path.exists()):
        v, lay = _detect(path)
        if layout not in ("auto", lay):
            raise ValueError(f"{path} is a {lay} store, not {layout}")
        cls = Plate if lay == "hcs" else Position
        return cls(path, v, "r" if mode == "r" else "a", channel_names)
    if version not in ("0.4", "0.5"):
        raise ValueError("version must be '0.4' or '0.5'")
    if layout not in ("hcs", "fov"):
        raise ValueError("layout must be 'hcs' or 'fov' when creating a store")
    if path.exists() and any(path.iterdir()):
        raise FileExistsError(f"{path} exists and is not empty (never overwritten, like the reference)")
    path.mkdir(parents=True, exist_ok=True)
    cls = Plate if layout == "hcs" else Position
    return cls(path, version, "w", channel_names)
