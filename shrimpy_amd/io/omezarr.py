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

Chunk codecs: raw (``bytes`` / no compressor), zlib / gzip, zstd, blosc 1.x frames, and the Zarr v3
``sharding_indexed`` codec with its CRC-32C protected index -- i.e. what the acquisition engine
writes (sharded blosc-zstd, ``shrimpy/mantis/mantis_engine.py:474-481``) is read and written here
(``io/codecs.py``; zstd itself comes from numcodecs / zstandard / pyarrow / the system libzstd,
whichever is present).  ``open_ome_zarr(prefer_iohub=True)`` returns iohub's own object when that
package is importable; ``as_volume_array`` gives either kind the whole-volume interface the CLI uses.
"""

from __future__ import annotations

import contextlib
import gzip
import itertools
import json
import os
import threading
import zlib

from pathlib import Path
from typing import Iterator, Sequence

import numpy as np

__all__ = ["open_ome_zarr", "Plate", "Position", "ZarrArray", "UnsupportedCodec", "as_volume_array",
           "create_level", "position_scale"]

AXES = [
    {"name": "T", "type": "time", "unit": "second"},
    {"name": "C", "type": "channel"},
    {"name": "Z", "type": "space", "unit": "micrometer"},
    {"name": "Y", "type": "space", "unit": "micrometer"},
    {"name": "X", "type": "space", "unit": "micrometer"},
]
_V3_DTYPES = {"float32": "<f4", "float64": "<f8", "uint16": "<u2", "uint8": "|u1", "int16": "<i2",
              "int32": "<i4", "uint32": "<u4"}


def host_cores() -> int:
    """Cores this process may use: its affinity mask, cut down to the cgroup CPU quota if one is set
    (a container with a 16-core share of a 256-core host reports 256 from ``os.cpu_count()``; more
    runnable threads than the share only buys CFS throttling)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def rank_cores() -> int:
    """This rank's share of ``host_cores()``: the ranks ``torch.distributed.run`` started on this host
    (``LOCAL_WORLD_SIZE``) read and write their own positions side by side, so each sizes its pools from an
    equal part of the box -- 8 ranks x (16 + 16) threads on one host was the first thing to break a node-wide
    store-to-store run."""
    try:
        local = int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1)
    except ValueError:
        local = 1
    return max(1, host_cores() // max(1, local))


# Threads one volume read / write fans its files out to.  ``None`` = min(16, rank_cores()).  (A streamed
# run reads unit k+1 while it writes unit k-1; the two pools block on the page cache and on zstd in turn,
# and measured best with the full share each -- profiles/r02_io_threads.jsonl.)
IO_THREADS = {"read": None, "write": None}


def io_thread_budget(read: int | None = None, write: int | None = None) -> dict:
    """Set (or with no arguments reset) the reader / writer thread budgets; returns the previous ones."""
    prev = dict(IO_THREADS)
    IO_THREADS["read"], IO_THREADS["write"] = read, write
    return prev


_TMPFS_MOUNTS: list[str] | None = None


def _on_tmpfs(path) -> bool:
    """Is ``path`` on a tmpfs mount (``/dev/shm``, a RAM-backed scratch)?  From /proc/mounts, longest prefix."""
    global _TMPFS_MOUNTS
    if _TMPFS_MOUNTS is None:
        mounts = []
        try:
            with open("/proc/mounts") as f:
                for line in f:
                    parts = line.split()
                    if len(parts) >= 3:
                        mounts.append((parts[1].replace("\\040", " "), parts[2] in ("tmpfs", "ramfs")))
        except OSError:
            pass
        _TMPFS_MOUNTS = sorted(mounts, key=lambda m: -len(m[0]))
    real = os.path.realpath(str(path))
    for mount, ram in _TMPFS_MOUNTS:
        if real == mount or real.startswith(mount.rstrip("/") + "/"):
            return ram
    return False


def _read_through_mapping(path) -> bool:
    """Copy chunk bytes out of a private read-only mapping instead of ``preadv``?  On tmpfs the FIRST read of a freshly
    written file through ``read`` runs at 11-15 GB/s however many threads share it (every page is moved to the active
    LRU list under one lock; a second read: 69 GB/s), through a mapping at 20 GB/s (``tools/probes/shard_read.py``,
    ``profiles/r05_shard_read.jsonl``; the streamed config-4 run: 0.064 -> 0.059 s per unit).  Used for the compressed
    chunks of ``read_volume_frames`` only: the 64 uncompressed chunk files of a volume, one reader thread each, are
    SLOWER through mappings (0.117 against 0.093 s per 2.15 GB volume).  On a real file system ``preadv`` stays: a
    mapping there turns an I/O error into SIGBUS.  ``LSR_READ_MMAP=0|1`` overrides."""
    env = os.environ.get("LSR_READ_MMAP")
    if env in ("0", "1"):
        return env == "1"
    return _on_tmpfs(path)


def _io_threads(role: str) -> int:
    env = os.environ.get("LSR_IO_THREADS")          # "r,w": measurement override
    if env:
        r, w = (int(v) for v in env.split(","))
        return max(1, r if role == "read" else w)
    n = IO_THREADS.get(role)
    return max(1, int(n)) if n else min(16, rank_cores())


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


class _BlockCodec:
    """bytes <-> one chunk-shaped block.  ``kind``: None (raw little-endian), ``gzip`` / ``zlib``,
    ``zstd`` (bare frames) or ``blosc`` (c-blosc 1.x frames, ``io/codecs.py``)."""

    def __init__(self, kind=None, **params):
        self.kind, self.params = kind, params

    # -- metadata ---------------------------------------------------------------------------
    @classmethod
    def from_v3(cls, codecs: list, where) -> "_BlockCodec":
        out = cls(None)
        for codec in codecs:
            name = codec["name"] if isinstance(codec, dict) else codec
            conf = codec.get("configuration", {}) if isinstance(codec, dict) else {}
            if name == "bytes":
                if conf.get("endian", "little") != "little":
                    raise UnsupportedCodec(f"big-endian chunks in {where}")
            elif name in ("gzip", "zlib"):
                out = cls(name, level=int(conf.get("level", 1)))
            elif name == "zstd":
                out = cls("zstd", level=int(conf.get("level", 1)))
            elif name == "blosc":
                shuffle = {"noshuffle": 0, "shuffle": 1, "bitshuffle": 2}.get(conf.get("shuffle", "shuffle"), 1)
                out = cls("blosc", cname=conf.get("cname", "zstd"), clevel=int(conf.get("clevel", 1)),
                          shuffle=shuffle, typesize=int(conf.get("typesize", 0)),
                          blocksize=int(conf.get("blocksize", 0)))
            elif name == "crc32c":
                out.params["crc32c"] = True
            else:
                raise UnsupportedCodec(
                    f"codec {name!r} in {where}: this reader handles bytes, gzip, zstd, blosc, crc32c and "
                    "sharding_indexed; install iohub (zarr/numcodecs) to read anything else")
        return out

    @classmethod
    def from_v2(cls, comp: dict | None, where) -> "_BlockCodec":
        if comp is None:
            return cls(None)
        cid = comp.get("id")
        if cid in ("zlib", "gzip"):
            return cls(cid, level=int(comp.get("level", 1)))
        if cid == "zstd":
            return cls("zstd", level=int(comp.get("level", 1)))
        if cid == "blosc":
            return cls("blosc", cname=comp.get("cname", "lz4"), clevel=int(comp.get("clevel", 5)),
                       shuffle=int(comp.get("shuffle", 1)) if int(comp.get("shuffle", 1)) >= 0 else 1,
                       typesize=0, blocksize=int(comp.get("blocksize", 0)))
        raise UnsupportedCodec(f"compressor {cid!r} in {where}: this reader handles zlib, gzip, zstd and blosc; "
                               "install iohub (zarr/numcodecs) to read anything else")

    def to_v3(self) -> list:
        codecs = [{"name": "bytes", "configuration": {"endian": "little"}}]
        if self.kind in ("gzip", "zlib"):
            codecs.append({"name": "gzip", "configuration": {"level": self.params.get("level", 1)}})
        elif self.kind == "zstd":
            codecs.append({"name": "zstd", "configuration": {"level": self.params.get("level", 1),
                                                             "checksum": False}})
        elif self.kind == "blosc":
            codecs.append({"name": "blosc", "configuration": {
                "cname": self.params["cname"], "clevel": self.params["clevel"],
                "shuffle": ["noshuffle", "shuffle", "bitshuffle"][self.params["shuffle"]],
                "typesize": self.params["typesize"], "blocksize": self.params.get("blocksize", 0)}})
        return codecs

    def to_v2(self) -> dict | None:
        if self.kind in ("gzip", "zlib"):
            return {"id": "zlib", "level": self.params.get("level", 1)}
        if self.kind == "zstd":
            return {"id": "zstd", "level": self.params.get("level", 1)}
        if self.kind == "blosc":
            return {"id": "blosc", "cname": self.params["cname"], "clevel": self.params["clevel"],
                    "shuffle": self.params["shuffle"], "blocksize": self.params.get("blocksize", 0)}
        return None

    @classmethod
    def named(cls, name: str | None, dtype: np.dtype, blocksize: int = 0) -> "_BlockCodec":
        """``None`` | "gzip" | "zlib" | "zstd" | "blosc-zstd" (the acquisition's) | "blosc-lz4".  ``blocksize``: bytes
        per blosc block (0: this package's default of 256 KB; c-blosc itself picks 32 KB for zstd at level 1, which is
        what the acquisition's frames have)."""
        if name in (None, "", "raw", "none"):
            return cls(None)
        if name in ("gzip", "zlib"):
            return cls(name, level=1)
        if name == "zstd":
            return cls("zstd", level=1)
        if name.startswith("blosc"):
            cname = name.split("-", 1)[1] if "-" in name else "zstd"
            return cls("blosc", cname=cname, clevel=1, shuffle=1, typesize=int(np.dtype(dtype).itemsize),
                       blocksize=int(blocksize))
        raise ValueError(f"unknown compression {name!r}")

    # -- data -------------------------------------------------------------------------------
    def decode(self, raw, shape, dtype, out: np.ndarray | None = None) -> np.ndarray:
        """Decode one block; into ``out`` (C-contiguous, chunk-shaped) when given."""
        from . import codecs

        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        if self.params.get("crc32c"):
            raw = memoryview(raw)
            if len(raw) < 4:
                raise ValueError("corrupt chunk: shorter than its CRC-32C suffix")
            want = int.from_bytes(raw[-4:], "little")
            raw = raw[:-4]
            if int(codecs.crc32c(raw)) != want:
                raise ValueError("corrupt chunk: CRC-32C mismatch")
        if self.kind == "blosc":
            if out is not None and out.flags.c_contiguous and out.dtype == dtype and out.nbytes == nbytes:
                codecs.blosc_decode(raw, out=out)
                return out
            data = codecs.blosc_decode(raw)
        elif self.kind == "gzip":
            data = gzip.decompress(bytes(raw))
        elif self.kind == "zlib":
            data = zlib.decompress(bytes(raw))
        elif self.kind == "zstd":
            data = codecs.zstd_decompress(raw, nbytes)
        else:
            data = raw
        block = np.frombuffer(data, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        if out is not None:
            out[...] = block
            return out
        return block

    def encode(self, block: np.ndarray) -> bytes:
        from . import codecs

        block = np.ascontiguousarray(block)
        if self.kind == "blosc":
            raw = codecs.blosc_encode(block, self.params.get("typesize") or block.dtype.itemsize,
                                      self.params["cname"], self.params["clevel"], self.params["shuffle"],
                                      self.params.get("blocksize", 0))
        elif self.kind == "gzip":
            raw = gzip.compress(block.tobytes(), compresslevel=self.params.get("level", 1))
        elif self.kind == "zlib":
            raw = zlib.compress(block.tobytes(), self.params.get("level", 1))
        elif self.kind == "zstd":
            raw = codecs.zstd_compress(block.tobytes(), self.params.get("level", 1))
        else:
            raw = block.tobytes()
        if self.params.get("crc32c"):
            raw += int(codecs.crc32c(raw)).to_bytes(4, "little")
        return raw


_MISSING = 0xFFFFFFFFFFFFFFFF   # (offset, nbytes) of an absent inner chunk in a shard index

_shard_locks_guard = threading.Lock()
_shard_locks: dict[str, list] = {}     # directory -> [lock, threads that hold or wait for it]


@contextlib.contextmanager
def _shard_lock(path: Path):
    """Exclusive access to one shard file for a read-modify-write: a lock per directory for the threads of this
    process, ``flock`` on the shard's DIRECTORY for other processes (ranks) on the same host.  The directory, not a
    ``<shard>.lock`` beside the chunk (nothing but chunk files lives in an array's key space), and not the shard itself (it is
    replaced by rename: a lock on the old inode would not hold anyone who opens the new one).

    Scope: one host.  ``flock`` does not reach ranks on other nodes, and some network file systems refuse it on a
    read-only directory descriptor (EBADF / ENOLCK where flock is emulated by byte-range locks): the in-process lock then
    still holds, with one warning -- a multi-node run must give every shard that spans several (t, c) volumes to ONE rank
    (``pipeline.run_sharded`` deals whole volumes; keep ``shards[:2] == (1, 1)``, what the acquisition writes, or shard
    the units by shard)."""
    folder = path.parent
    key = str(folder)
    # An entry is counted while a thread holds OR waits for its lock and is dropped by the last one to leave: an entry
    # can never be purged between a thread fetching its lock and acquiring it (which would let a second Lock for the same
    # directory into the table and two threads into one shard's read-modify-write), and a long plate run, which visits
    # every array directory once, does not keep them all.
    with _shard_locks_guard:
        entry = _shard_locks.setdefault(key, [threading.Lock(), 0])
        entry[1] += 1
    try:
        with entry[0]:
            with _flocked(folder):
                yield
    finally:
        with _shard_locks_guard:
            entry[1] -= 1
            if entry[1] == 0 and _shard_locks.get(key) is entry:
                del _shard_locks[key]


@contextlib.contextmanager
def _flocked(folder: Path):
    """``flock`` on the directory (other processes on this host); a refusal is reported once and tolerated."""
    import fcntl

    fd = None
    try:
        fd = os.open(str(folder), os.O_RDONLY)
        fcntl.flock(fd, fcntl.LOCK_EX)
    except OSError as exc:
        if fd is not None:
            os.close(fd)
            fd = None
        global _flock_warned
        if not _flock_warned:
            _flock_warned = True
            import warnings

            warnings.warn(f"flock on {folder} failed ({exc}); shards that span several volumes are protected against "
                          "this process's threads only", RuntimeWarning, stacklevel=4)
    try:
        yield
    finally:
        if fd is not None:
            try:
                fcntl.flock(fd, fcntl.LOCK_UN)
            finally:
                os.close(fd)


_flock_warned = False


class ZarrArray:
    """One N-D array; whole ``(t, c)`` volumes are the unit of I/O.

    ``chunks`` is the shape of one encoded block; with the Zarr v3 ``sharding_indexed`` codec
    (what the acquisition writes, ``shrimpy/mantis/mantis_engine.py:474-481``) several blocks share
    one file of shape ``shards`` and an index of (offset, nbytes) pairs, CRC-32C protected, at its end
    or start.  Without sharding ``shards`` is ``None`` and every block is its own file.
    """

    def __init__(self, path: Path, version: str, mode: str):
        self.path, self.version, self.mode = Path(path), version, mode
        self.shards = None
        self._index_location, self._index_crc = "end", True
        if version == "0.5":
            meta = _read_json(self.path / "zarr.json")
            self.shape = tuple(meta["shape"])
            grid = tuple(meta["chunk_grid"]["configuration"]["chunk_shape"])
            self.dtype = np.dtype(_V3_DTYPES.get(meta["data_type"], meta["data_type"]))
            self.fill_value = meta.get("fill_value", 0) or 0
            enc = meta.get("chunk_key_encoding", {"name": "default"})
            self._sep = enc.get("configuration", {}).get("separator", "/" if enc.get("name", "default") == "default" else ".")
            self._prefix = "c" + self._sep if enc.get("name", "default") == "default" else ""
            codecs = meta.get("codecs", [])
            names = [c["name"] if isinstance(c, dict) else c for c in codecs]
            if "sharding_indexed" in names:
                if names != ["sharding_indexed"]:
                    raise UnsupportedCodec(f"codecs {names} around sharding_indexed in {self.path}")
                conf = codecs[0]["configuration"]
                self.shards = grid
                self.chunks = tuple(conf["chunk_shape"])
                if len(self.chunks) != len(grid) or any(s % c for s, c in zip(grid, self.chunks)):
                    raise UnsupportedCodec(f"shard shape {grid} is not a multiple of its chunk shape {self.chunks}")
                self._index_location = conf.get("index_location", "end")
                inames = [c["name"] if isinstance(c, dict) else c for c in conf.get("index_codecs", [])]
                if [n for n in inames if n not in ("bytes", "crc32c")]:
                    raise UnsupportedCodec(f"shard index codecs {inames} in {self.path} (bytes / crc32c are handled)")
                self._index_crc = "crc32c" in inames
                self._codec = _BlockCodec.from_v3(conf.get("codecs", []), self.path)
            else:
                self.chunks = grid
                self._codec = _BlockCodec.from_v3(codecs, self.path)
        else:
            meta = _read_json(self.path / ".zarray")
            self.shape = tuple(meta["shape"])
            self.chunks = tuple(meta["chunks"])
            self.dtype = np.dtype(meta["dtype"])
            self.fill_value = meta.get("fill_value", 0) or 0
            self._sep = meta.get("dimension_separator", ".")
            self._prefix = ""
            self._codec = _BlockCodec.from_v2(meta.get("compressor"), self.path)
            if meta.get("order", "C") != "C" or meta.get("filters"):
                raise UnsupportedCodec("only C-order arrays without filters are supported")
        if self._codec.kind == "blosc" and not self._codec.params.get("typesize"):
            self._codec.params["typesize"] = self.dtype.itemsize
        self._compress = self._codec.kind   # (kept: "is this array stored raw?")

    # -- creation ---------------------------------------------------------------------------
    @classmethod
    def create(cls, path: Path, version: str, shape, chunks, dtype, compress: str | None = None,
               shards=None, blocksize: int = 0):
        """``compress``: see ``_BlockCodec.named``.  ``shards`` (NGFF 0.5 only): shape of the files,
        a multiple of ``chunks`` -- ``compress="blosc-zstd"`` with shards is the acquisition's layout."""
        path = Path(path)
        dtype = np.dtype(dtype)
        codec = _BlockCodec.named(compress, dtype, blocksize)
        if version == "0.5":
            inner = codec.to_v3()
            if shards is not None:
                shards = tuple(int(s) for s in shards)
                if len(shards) != len(chunks) or any(s % c for s, c in zip(shards, chunks)):
                    raise ValueError(f"shards {shards} must be a multiple of chunks {tuple(chunks)}")
                codecs = [{"name": "sharding_indexed", "configuration": {
                    "chunk_shape": list(chunks), "codecs": inner,
                    "index_codecs": [{"name": "bytes", "configuration": {"endian": "little"}},
                                     {"name": "crc32c"}],
                    "index_location": "end"}}]
                grid = list(shards)
            else:
                codecs, grid = inner, list(chunks)
            _write_json(path / "zarr.json", {
                "zarr_format": 3, "node_type": "array", "shape": list(shape), "data_type": dtype.name,
                "chunk_grid": {"name": "regular", "configuration": {"chunk_shape": grid}},
                "chunk_key_encoding": {"name": "default", "configuration": {"separator": "/"}},
                "fill_value": 0, "codecs": codecs,
                "dimension_names": [a["name"] for a in AXES][-len(shape):],
            })
        else:
            if shards is not None:
                raise ValueError("sharding needs NGFF 0.5 (Zarr v3)")
            _write_json(path / ".zarray", {
                "zarr_format": 2, "shape": list(shape), "chunks": list(chunks), "dtype": dtype.str,
                "compressor": codec.to_v2(), "fill_value": 0,
                "order": "C", "filters": None, "dimension_separator": "/",
            })
        return cls(path, version, "w")

    # -- file / block addressing ------------------------------------------------------------
    def _file_path(self, idx: Sequence[int]) -> Path:
        return self.path / (self._prefix + self._sep.join(str(i) for i in idx))

    _chunk_path = _file_path   # (unsharded arrays: one file per chunk)

    def _read_chunk(self, idx) -> np.ndarray | None:
        p = self._file_path(idx)
        if not p.exists():
            return None
        return self._codec.decode(p.read_bytes(), self.chunks, self.dtype)

    def _write_chunk(self, idx, block: np.ndarray) -> None:
        p = self._file_path(idx)
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_bytes(self._codec.encode(np.ascontiguousarray(block, dtype=self.dtype)))

    def _check_lead(self, lead) -> tuple[int, ...]:
        k = len(self.shape) - 3
        if len(lead) != k:
            raise IndexError(f"expected {k} leading indices, got {len(lead)}")
        for i, n in zip(lead, self.shape):
            if not 0 <= i < n:
                raise IndexError(f"index {tuple(lead)} out of range for shape {self.shape}")
        for c in self.chunks[:k]:
            if c != 1:
                raise UnsupportedCodec("leading (T, C) chunk sizes other than 1 are not supported")
        return tuple(int(i) for i in lead)

    def _grid(self, lead: tuple[int, ...]):
        """File indices (trailing three dims) covering the (Z, Y, X) volume at ``lead``."""
        k = len(lead)
        files = self.shards or self.chunks
        ranges = [range(-(-self.shape[d] // files[d])) for d in range(k, len(self.shape))]
        return itertools.product(*ranges)

    # Files of one volume are independent: they are read / written by a small thread pool (file
    # I/O, zlib and the C codecs release the GIL), and a block that is a whole contiguous z-range of
    # the volume -- the layout the acquisition writes, chunks (1, 1, <=32, ny, nx),
    # ``shrimpy/dynatrack/tracking.py:1337-1367`` -- moves between the file and the caller's buffer
    # (e.g. a pinned staging slot) without an intermediate copy.
    _POOL_MIN_BYTES = 8 << 20

    def _map_chunks(self, fn, cidxs, nbytes, role="read"):
        workers = min(_io_threads(role), len(cidxs))
        if workers <= 1 or nbytes < self._POOL_MIN_BYTES:
            for c in cidxs:
                fn(c)
            return
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(workers, "lsr-zarr") as pool:
            list(pool.map(fn, cidxs))  # list(): re-raise the first worker exception

    def _inner_threads(self, n_files: int, nbytes: int, role: str) -> int:
        """Threads ONE shard may spread its inner chunks over: the budget left by the fan-out over
        files (a volume stored as a single shard -- the acquisition's layout -- would otherwise be
        decoded by one thread, chunk after chunk)."""
        if nbytes < self._POOL_MIN_BYTES:
            return 1
        budget = _io_threads(role)
        return max(1, budget // max(1, min(budget, n_files)))

    @staticmethod
    def _fan_out(fn, items, threads: int) -> list:
        items = list(items)
        if threads <= 1 or len(items) <= 1:
            return [fn(i) for i in items]
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(min(threads, len(items)), "lsr-shard") as pool:
            return list(pool.map(fn, items))

    def _slab(self, vol: np.ndarray, sl, block_shape) -> np.ndarray | None:
        """``vol[sl]`` when that is a whole block and contiguous in ``vol`` (decode / encode in place)."""
        full = all(s.stop - s.start == c for s, c in zip(sl, block_shape))
        if not full or vol.dtype != self.dtype or not vol.flags.c_contiguous:
            return None
        if (sl[1].start, sl[1].stop, sl[2].start, sl[2].stop) != (0, vol.shape[1], 0, vol.shape[2]):
            return None
        return vol[sl]

    def _slab_view(self, vol: np.ndarray, sl) -> memoryview | None:
        view = self._slab(vol, sl, self.chunks[len(self.shape) - 3:]) if self._codec.kind is None else None
        return None if view is None else memoryview(view).cast("B")

    # -- shards -----------------------------------------------------------------------------
    def _shard_counts(self) -> tuple[int, ...]:
        return tuple(s // c for s, c in zip(self.shards, self.chunks))

    def _read_shard_index(self, f, size: int) -> np.ndarray:
        """(n_inner..., 2) uint64 array of (offset, nbytes) from an open shard file."""
        from .codecs import crc32c

        counts = self._shard_counts()
        n = int(np.prod(counts))
        ilen = 16 * n + (4 if self._index_crc else 0)
        if size < ilen:
            raise OSError(f"shard {f.name} is shorter than its index ({size} < {ilen} bytes)")
        f.seek(size - ilen if self._index_location == "end" else 0)
        raw = f.read(ilen)
        if self._index_crc:
            want = int.from_bytes(raw[-4:], "little")
            if crc32c(raw[:-4]) != want:
                raise OSError(f"shard {f.name}: index checksum mismatch (file truncated or corrupt)")
            raw = raw[:-4]
        return np.frombuffer(raw, dtype="<u8").reshape(counts + (2,))

    def _inner_of(self, lead, fidx):
        """Inner-chunk coordinates of shard ``fidx`` that belong to the volume at ``lead``:
        yields (index into the shard's chunk grid, slices into the volume)."""
        k = len(lead)
        counts = self._shard_counts()
        vshape, vchunks, vshard = self.shape[k:], self.chunks[k:], self.shards[k:]
        head = tuple(i % s for i, s in zip(lead, self.shards[:k]))
        for inner in itertools.product(*[range(c) for c in counts[k:]]):
            lo = [f * s + i * c for f, s, i, c in zip(fidx, vshard, inner, vchunks)]
            if any(a >= n for a, n in zip(lo, vshape)):
                continue
            yield head + inner, tuple(slice(a, min(a + c, n)) for a, c, n in zip(lo, vchunks, vshape))

    def _read_shard(self, lead, fidx, out: np.ndarray, threads: int = 1) -> None:
        k = len(lead)
        vchunks = self.chunks[k:]
        path = self._file_path(tuple(i // s for i, s in zip(lead, self.shards[:k])) + tuple(fidx))
        try:
            f = open(path, "rb", buffering=0)
        except FileNotFoundError:
            for _, sl in self._inner_of(lead, fidx):
                out[sl] = self.fill_value
            return
        with f:
            fd = f.fileno()
            size = os.fstat(fd).st_size
            index = self._read_shard_index(f, size)

            def one(item):
                inner, sl = item
                off, nb = (int(v) for v in index[inner])
                if off == _MISSING and nb == _MISSING:
                    out[sl] = self.fill_value
                    return
                if off + nb > size:
                    raise OSError(f"shard {path}: chunk {inner} runs past the end of the file")
                raw = os.pread(fd, nb, off)          # positional: the workers share the descriptor
                if len(raw) != nb:
                    raise OSError(f"shard {path}: short read of chunk {inner}")
                dest = self._slab(out, sl, vchunks)
                if dest is not None:
                    self._codec.decode(raw, dest.shape, self.dtype, out=dest)
                else:
                    block = self._codec.decode(raw, vchunks, self.dtype)
                    out[sl] = block[tuple(slice(0, s.stop - s.start) for s in sl)]

            self._fan_out(one, self._inner_of(lead, fidx), threads)

    def _write_shard(self, lead, fidx, vol, threads: int = 1, encoded: dict | None = None) -> None:
        k = len(lead)
        path = self._file_path(tuple(i // s for i, s in zip(lead, self.shards[:k])) + tuple(fidx))
        if any(s > 1 for s in self.shards[:k]):
            # the shard also holds other (t, c) volumes: a read-modify-write, one writer at a time --
            # writer threads of a streamed run, or ranks, may hold different volumes of this shard
            path.parent.mkdir(parents=True, exist_ok=True)
            with _shard_lock(path):
                self._write_shard_locked(path, lead, fidx, vol, threads, merge=True, encoded=encoded)
        else:
            self._write_shard_locked(path, lead, fidx, vol, threads, merge=False, encoded=encoded)

    def _write_shard_locked(self, path, lead, fidx, vol, threads, merge: bool, encoded: dict | None = None) -> None:
        from .codecs import crc32c

        k = len(lead)
        vchunks = self.chunks[k:]
        counts = self._shard_counts()
        blobs: dict[tuple, bytes] = {}
        if merge and path.exists():
            # the shard also holds other (t, c) volumes: keep their encoded chunks as they are
            with open(path, "rb", buffering=0) as f:
                size = os.fstat(f.fileno()).st_size
                old = self._read_shard_index(f, size)
                for inner in itertools.product(*[range(c) for c in counts]):
                    off, nb = (int(v) for v in old[inner])
                    if off != _MISSING:
                        f.seek(off)
                        blobs[inner] = f.read(nb)
        def encode(item):
            inner, sl = item
            block = self._slab(vol, sl, vchunks)
            if block is None:
                block = np.zeros(vchunks, dtype=self.dtype)
                block[tuple(slice(0, s.stop - s.start) for s in sl)] = vol[sl]
            return inner, self._codec.encode(np.ascontiguousarray(block, dtype=self.dtype))

        if encoded is not None:        # chunks that arrive as frames (written on the device): stored as they are
            blobs.update(encoded)
        else:
            blobs.update(self._fan_out(encode, self._inner_of(lead, fidx), threads))
        index = np.full(counts + (2,), _MISSING, dtype="<u8")
        ilen = index.nbytes + (4 if self._index_crc else 0)
        pos = ilen if self._index_location == "start" else 0
        order = sorted(blobs)
        for inner in order:
            index[inner] = (pos, len(blobs[inner]))
            pos += len(blobs[inner])
        ibytes = index.tobytes()
        if self._index_crc:
            ibytes += int(crc32c(ibytes)).to_bytes(4, "little")
        path.parent.mkdir(parents=True, exist_ok=True)
        tmp = path.with_name(f"{path.name}.{os.getpid()}.{threading.get_ident()}.partial")   # never shared
        with open(tmp, "wb") as f:
            if self._index_location == "start":
                f.write(ibytes)
            for inner in order:
                f.write(blobs[inner])
            if self._index_location != "start":
                f.write(ibytes)
        os.replace(tmp, path)   # a killed run never leaves a shard with a stale index behind

    # -- volumes ----------------------------------------------------------------------------
    def read_volume(self, *lead: int, out: np.ndarray | None = None) -> np.ndarray:
        """The (Z, Y, X) volume at leading indices (t, c) as a C-contiguous array; with ``out``
        (same shape and dtype, e.g. a pinned staging buffer) the chunks are decoded into it."""
        lead = self._check_lead(lead)
        k = len(lead)
        vshape, vchunks = self.shape[k:], self.chunks[k:]
        if out is None:
            out = np.empty(vshape, dtype=self.dtype)
        elif tuple(out.shape) != tuple(vshape) or out.dtype != self.dtype:
            raise ValueError(f"out must be {tuple(vshape)} {self.dtype}, got {out.shape} {out.dtype}")

        files = list(self._grid(lead))
        inner_threads = self._inner_threads(len(files), out.nbytes, "read") if self.shards is not None else 1

        def read_one(cidx):
            if self.shards is not None:
                self._read_shard(lead, cidx, out, inner_threads)
                return
            sl = tuple(slice(c * s, min((c + 1) * s, n)) for c, s, n in zip(cidx, vchunks, vshape))
            path = self._file_path(lead + cidx)
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
            try:
                raw = path.read_bytes()
            except FileNotFoundError:
                out[sl] = self.fill_value
                return
            dest = self._slab(out, sl, vchunks)
            if dest is not None:
                self._codec.decode(raw, dest.shape, self.dtype, out=dest)
                return
            block = self._codec.decode(raw, vchunks, self.dtype)
            out[sl] = block[tuple(slice(0, s.stop - s.start) for s in sl)]

        self._map_chunks(read_one, files, out.nbytes)
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
        lead = self._check_lead(lead)
        vchunks = self.chunks[k:]

        files = list(self._grid(lead))
        inner_threads = self._inner_threads(len(files), vol.nbytes, "write") if self.shards is not None else 1

        def write_one(cidx):
            if self.shards is not None:
                self._write_shard(lead, cidx, vol, inner_threads)
                return
            sl = tuple(slice(c * s, min((c + 1) * s, n)) for c, s, n in zip(cidx, vchunks, vol.shape))
            view = self._slab_view(vol, sl)
            if view is not None:
                path = self._file_path(lead + cidx)
                path.parent.mkdir(parents=True, exist_ok=True)
                with open(path, "wb", buffering=0) as f:
                    done = 0
                    while done < len(view):
                        done += f.write(view[done:])
                return
            block = self._slab(vol, sl, vchunks)
            if block is None:
                block = np.zeros(self.chunks[k:], dtype=self.dtype)
                block[tuple(slice(0, s.stop - s.start) for s in sl)] = vol[sl]
            self._write_chunk(lead + cidx, block)

        self._map_chunks(write_one, files, vol.nbytes, "write")

    # -- volumes as they lie in the store (for the device-side decoder) -----------------------
    def _frame_sources(self, lead):
        """[(z-chunk index, path, offset, nbytes)] of every stored chunk of the volume at ``lead`` (absent chunks are
        left out); offset / nbytes are ``None`` for a chunk that is a whole file."""
        k = len(lead)
        zc = self.chunks[k]
        nz = -(-self.shape[k] // zc)
        out = []
        if self.shards is None:
            for i in range(nz):
                path = self._file_path(lead + (i, 0, 0))
                try:
                    out.append((i, path, 0, os.stat(path).st_size))
                except FileNotFoundError:
                    pass
            return out
        per = self.shards[k] // zc
        head = tuple(i % s for i, s in zip(lead, self.shards[:k]))
        for fz in range(-(-nz // per)):
            path = self._file_path(tuple(i // s for i, s in zip(lead, self.shards[:k])) + (fz, 0, 0))
            try:
                with open(path, "rb", buffering=0) as f:
                    size = os.fstat(f.fileno()).st_size
                    index = self._read_shard_index(f, size)
            except FileNotFoundError:
                continue
            for i in range(fz * per, min((fz + 1) * per, nz)):
                off, nb = (int(v) for v in index[head + (i - fz * per, 0, 0)])
                if off == _MISSING and nb == _MISSING:
                    continue
                if off + nb > size:
                    raise OSError(f"shard {path}: chunk {i} runs past the end of the file")
                out.append((i, path, off, nb))
        return out

    def compressed_layout(self, *lead) -> dict | None:
        """What the device-side decoder (``io/device_codec.DeviceBloscDecoder``) needs to take this array's volumes as
        they lie in the store -- ``dict(nbytes, blocksize, typesize, n_frames)`` read from the first stored chunk of the
        volume at ``lead`` -- or ``None`` when the chunks are not blosc-zstd frames of whole (Y, X) planes."""
        from .device_codec import frame_layout

        k = len(self.shape) - 3
        if self._codec.kind != "blosc" or self.dtype.itemsize not in (1, 2, 4):
            return None
        if any(v != 1 for v in self.chunks[:k]) or tuple(self.chunks[k + 1:]) != tuple(self.shape[k + 1:]):
            return None
        if self.shards is not None and tuple(self.shards[k + 1:]) != tuple(self.shape[k + 1:]):
            return None
        lead = self._check_lead(lead)
        want = int(np.prod(self.chunks[k:])) * self.dtype.itemsize
        for _, path, off, nb in self._frame_sources(lead):
            with open(path, "rb", buffering=0) as f:
                head = os.pread(f.fileno(), 16, off)
            lay = frame_layout(head)
            if lay is None or lay["nbytes"] != want or lay["typesize"] != self.dtype.itemsize:
                return None
            lay["n_frames"] = -(-self.shape[k] // self.chunks[k])
            return lay
        return None

    def read_volume_frames(self, *lead, out: np.ndarray):
        """The volume at ``lead`` as its stored chunk frames: the compressed bytes are read one after the other into
        ``out`` (uint8, e.g. a pinned staging slot) and a ``device_codec.CompressedVolume`` says where each z-chunk's
        frame lies (absent chunks: size 0).  No entropy decoding happens on the host; a CRC-32C suffix, where the codec
        chain has one, is checked and left out."""
        from .codecs import crc32c
        from .device_codec import CompressedVolume

        lead = self._check_lead(lead)
        k = len(lead)
        nz = -(-self.shape[k] // self.chunks[k])
        crc = bool(self._codec.params.get("crc32c"))
        buf = out.reshape(-1).view(np.uint8)
        sources = self._frame_sources(lead)
        table = np.zeros((nz, 2), dtype=np.int64)
        at = 0
        for i, _, _, nb in sources:
            table[i] = (at, nb - (4 if crc else 0))
            at += (nb + 15) // 16 * 16
        if at > buf.size:
            raise ValueError(f"the compressed chunks take {at} bytes, the buffer holds {buf.size}")

        import mmap

        maps: dict = {}
        maps_lock = threading.Lock()

        def mapping(path):
            with maps_lock:
                m = maps.get(path)
                if m is None:
                    with open(path, "rb") as f:
                        m = maps[path] = np.frombuffer(mmap.mmap(f.fileno(), 0, prot=mmap.PROT_READ), dtype=np.uint8)
                return m

        mapped = bool(sources) and _read_through_mapping(sources[0][1])

        def read_one(item):
            i, path, off, nb = item
            dest = memoryview(buf[int(table[i, 0]):int(table[i, 0]) + nb])
            if mapped:
                src = mapping(path)
                if off + nb > src.size:
                    raise OSError(f"{path}: chunk {i} runs past the end of the file")
                np.copyto(buf[int(table[i, 0]):int(table[i, 0]) + nb], src[off:off + nb])     # (GIL released)
            else:
                with open(path, "rb", buffering=0) as f:
                    got = 0
                    while got < nb:
                        n = os.preadv(f.fileno(), [dest[got:]], off + got)
                        if n <= 0:
                            raise OSError(f"{path}: short read of chunk {i}")
                        got += n
            if crc:
                if nb < 4 or int(crc32c(dest[:nb - 4])) != int.from_bytes(dest[nb - 4:nb], "little"):
                    raise ValueError(f"{path}: chunk {i}: CRC-32C mismatch")

        self._map_chunks(read_one, sources, at, "read")
        return CompressedVolume(table, at)

    def encoded_frame_bytes(self) -> int | None:
        """Decoded bytes of one chunk when this array can store chunks that arrive as blosc-zstd frames written on
        the device (``io/device_codec.py``): blosc / zstd / byte shuffle of the array's own element size, chunks that
        are whole (Y, X) planes -- else ``None`` (the volume is then encoded on the host, chunk by chunk)."""
        k = len(self.shape) - 3
        c = self._codec
        if c.kind != "blosc" or c.params.get("cname") != "zstd" or self.dtype.itemsize not in (1, 2, 4):
            return None
        if (c.params.get("typesize") or self.dtype.itemsize) != self.dtype.itemsize:
            return None
        if self.dtype.itemsize > 1 and c.params.get("shuffle", 1) != 1:     # (one-byte elements: nothing to shuffle)
            return None
        if any(v != 1 for v in self.chunks[:k]) or tuple(self.chunks[k + 1:]) != tuple(self.shape[k + 1:]):
            return None
        if self.shards is not None and tuple(self.shards[k + 1:]) != tuple(self.shape[k + 1:]):
            return None
        return int(np.prod(self.chunks[k:])) * self.dtype.itemsize

    def write_encoded_volume(self, *args) -> None:
        """``write_encoded_volume(t, c, frames)``: store one volume whose chunks are already blosc frames
        (``frames[i]`` = z-chunk i, each decoding to ``encoded_frame_bytes()`` bytes, the last one zero-padded) --
        what ``staging.EncodedVolume.frames`` holds.  The host only writes bytes (plus the CRC-32C suffix where the
        store's codec chain asks for one)."""
        if self.mode == "r":
            raise PermissionError("store opened read-only")
        *lead, frames = args
        lead = self._check_lead(lead)
        k = len(lead)
        want = self.encoded_frame_bytes()
        if want is None:
            raise UnsupportedCodec(f"{self.path}: chunks of this array cannot be stored as device-written blosc-zstd frames")
        zc = self.chunks[k]
        nz = -(-self.shape[k] // zc)
        if len(frames) != nz:
            raise ValueError(f"expected {nz} frames (z chunks of {zc} planes), got {len(frames)}")
        from . import codecs

        crc = bool(self._codec.params.get("crc32c"))

        def blob(i):
            f = frames[i]
            head = codecs.blosc_header(bytes(f[:16]))
            if head["nbytes"] != want or head["cbytes"] != len(f):
                raise ValueError(f"frame {i}: holds {head['nbytes']} bytes in {head['cbytes']}, expected {want} in {len(f)}")
            return f if not crc else bytes(f) + int(codecs.crc32c(f)).to_bytes(4, "little")

        if self.shards is not None:
            per = self.shards[k] // zc
            counts = self._shard_counts()
            head = tuple(i % s for i, s in zip(lead, self.shards[:k]))

            def write_shard(fz):
                enc = {head + (i - fz * per,) + (0,) * (len(counts) - k - 1): bytes(blob(i))
                       for i in range(fz * per, min((fz + 1) * per, nz))}
                self._write_shard(lead, (fz, 0, 0), None, 1, encoded=enc)

            self._map_chunks(write_shard, list(range(-(-nz // per))), sum(len(f) for f in frames), "write")
            return

        def write_one(i):
            path = self._file_path(lead + (i, 0, 0))
            path.parent.mkdir(parents=True, exist_ok=True)
            view = memoryview(blob(i))
            with open(path, "wb", buffering=0) as f:
                done = 0
                while done < len(view):
                    done += f.write(view[done:])

        self._map_chunks(write_one, list(range(nz)), sum(len(f) for f in frames), "write")

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
                     compress: str | None = None, shards=None, blocksize: int = 0, translation=None) -> ZarrArray:
        """Create level ``name`` (TCZYX).  Default chunks follow the reference:
        ``(1, 1, min(32, nz), ny, nx)`` (``shrimpy/dynatrack/tracking.py:1362``).
        ``compress="blosc-zstd"`` with ``shards="volume"`` (one shard file per (t, c) volume) or an
        explicit shard shape reproduces the acquisition's layout
        (``shrimpy/mantis/mantis_engine.py:474-481``, NGFF 0.5 only)."""
        shape = tuple(int(s) for s in shape)
        if len(shape) != 5:
            raise ValueError(f"expected a 5-D TCZYX shape, got {shape}")
        if chunks is None:
            chunks = (1, 1, min(32, shape[2]), shape[3], shape[4])
        if isinstance(shards, str):
            if shards != "volume":
                raise ValueError("shards must be a shape or 'volume'")
            shards = (1, 1) + tuple(-(-n // c) * c for n, c in zip(shape[2:], chunks[2:]))
        arr = ZarrArray.create(self.path / name, self.version, shape, chunks, dtype, compress, shards, blocksize)
        scale = [float(s) for s in (scale if scale is not None else (1, 1, 1, 1, 1))]
        transforms = [{"type": "scale", "scale": scale}]
        if translation is not None and any(float(v) != 0.0 for v in translation):
            # NGFF: scale first, then translation, in physical units (where index 0 of this level sits)
            transforms.append({"type": "translation", "translation": [float(v) for v in translation]})
        attrs = {
            "multiscales": [{
                "version": self.version, "axes": AXES, "name": "0",
                "datasets": [{"path": name, "coordinateTransformations": transforms}],
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


# ---------------------------------------------------------------------------------------------
# One whole-volume interface over this module's arrays AND iohub's (``ImageArray`` is a zarr array:
# numpy indexing and assignment, ``shape`` / ``dtype`` / ``chunks``; read surface
# ``shrimpy/replay_camera.py:176-268``, write surface ``shrimpy/dynatrack/tracking.py:1337-1367``).
# ---------------------------------------------------------------------------------------------


class _IndexedVolumes:
    """``read_volume`` / ``write_volume`` on top of any array that indexes like numpy."""

    def __init__(self, array):
        self._a = array
        self.shape = tuple(int(n) for n in array.shape)
        self.dtype = np.dtype(array.dtype)
        self.chunks = tuple(getattr(array, "chunks", None) or self.shape)

    def _lead(self, lead):
        k = len(self.shape) - 3
        if len(lead) != k:
            raise IndexError(f"expected {k} leading indices, got {len(lead)}")
        for i, n in zip(lead, self.shape):
            if not 0 <= i < n:
                raise IndexError(f"index {tuple(lead)} out of range for shape {self.shape}")
        return tuple(int(i) for i in lead)

    def read_volume(self, *lead, out=None):
        vol = np.asarray(self._a[self._lead(lead)])
        if out is None:
            return np.ascontiguousarray(vol)
        if tuple(out.shape) != vol.shape or out.dtype != vol.dtype:
            raise ValueError(f"out must be {vol.shape} {vol.dtype}, got {out.shape} {out.dtype}")
        np.copyto(out, vol)
        return out

    def write_volume(self, *args):
        *lead, vol = args
        self._a[self._lead(lead)] = np.asarray(vol, dtype=self.dtype)


def as_volume_array(array):
    """``array`` itself when it already moves whole volumes (this module's ``ZarrArray``), else a
    wrapper that does so through numpy indexing (iohub / zarr / dask / numpy arrays)."""
    if hasattr(array, "read_volume") and hasattr(array, "write_volume"):
        return array
    return _IndexedVolumes(array)


def position_scale(position) -> tuple[float, ...]:
    """(T, C, Z, Y, X) scale of a position's level 0, from either kind of object -- the metadata walk
    of ``shrimpy/replay_camera.py:256-266``."""
    scale = getattr(position, "scale", None)
    if scale is not None and not callable(scale):
        return tuple(float(v) for v in scale)
    ms = position.zattrs.get("multiscales", [{}])
    ds = ms[0].get("datasets", [{}]) if ms else [{}]
    for t in ds[0].get("coordinateTransformations", []):
        if t.get("type") == "scale":
            return tuple(float(v) for v in t["scale"])
    return (1.0,) * 5


def create_level(position, shape, dtype, scale, chunks=None, name: str = "0", translation=None, **kw):
    """``position.create_zeros`` with the scale metadata, for this module's ``Position`` (``scale=``)
    and for iohub's (``transform=[TransformationMeta(type="scale", ...)]``,
    ``scripts/measure_psf.py:273-287``)."""
    shape = tuple(int(n) for n in shape)
    if chunks is None:
        # the reference's (1, 1, min(32, nz), ny, nx) (tracking.py:1362), with the z extent cut down
        # where that would make chunks of hundreds of MB: a deskewed (86, 2048, 2491) float32 volume
        # in 32-plane chunks is three 650 MB files, and buffered writes to ONE file serialise on its
        # inode lock (measured: 6 GB/s, 0.29 s per volume against 0.03 s of kernels); ~64 MB chunks
        # are 29 files that sixteen threads write side by side
        plane_bytes = shape[3] * shape[4] * np.dtype(dtype).itemsize
        zc = max(1, min(32, shape[2], (64 << 20) // max(plane_bytes, 1)))
        chunks = (1, 1, zc, shape[3], shape[4])
    if isinstance(position, Position):
        return position.create_zeros(name, shape=shape, dtype=dtype, chunks=chunks, scale=scale, translation=translation, **kw)
    from iohub.ngff.models import TransformationMeta

    transform = [TransformationMeta(type="scale", scale=[float(v) for v in scale])]
    if translation is not None and any(float(v) != 0.0 for v in translation):
        transform.append(TransformationMeta(type="translation", translation=[float(v) for v in translation]))
    return position.create_zeros(name, shape=shape, dtype=dtype, chunks=chunks, transform=transform)
