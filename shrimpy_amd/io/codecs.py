"""Chunk codecs of the Zarr stores on either side of the hot path: blosc, zstd, lz4, gzip, crc32c.

The acquisition engine writes Zarr v3 shards whose inner chunks are blosc frames with the zstd
compressor (``shrimpy/mantis/mantis_engine.py:474-481``: ``compression="blosc-zstd"``; asserted in
``shrimpy/tests/test_mantis_integration.py:177-190``).  Reading that needs three things a plain
Python install lacks -- the sharding index checksum (CRC-32C), the blosc 1.x frame and zstd -- so
this module implements the first two and borrows the third from whatever the host has:

* ``crc32c``            -- table-driven, pure Python/numpy (the index is a few hundred bytes);
* blosc 1.x frames      -- ``blosc_decode`` / ``blosc_encode`` follow the published c-blosc 1.x
  layout (16-byte header, ``bstarts`` table, per-block streams with the typesize split rule, byte-
  and bit-shuffle).  Who decodes, in order (``blosc_backend()``): a ``libblosc`` when one is
  loadable (``LSR_LIBBLOSC`` or the system one) -- same bytes, C speed, straight into the caller's
  buffer; ``numcodecs.blosc`` where that package is installed; this package's own frame walker in
  liblsrecon (``lsr_blosc_decode_host``, ``csrc/blosc_frame.hip``: host C++, the system zstd /
  lz4 / zlib by ``dlopen``, one call per chunk with the GIL released -- as fast as libblosc, and what
  a host without either library uses); the pure-Python walker last (it holds the GIL for ~10 us
  per stream, 20x slower on the acquisition's 32 KB blocks however many threads decode) and as the
  cross-check of the native one;
* zstd / lz4 block codecs -- first hit of: ``numcodecs``, ``zstandard`` / ``lz4``, ``pyarrow``, the
  system ``libzstd`` / ``liblz4`` through ctypes.

Nothing here touches the GPU; it is the data format either side of the kernels.
"""

from __future__ import annotations

import ctypes
import ctypes.util
import os
import struct
import sys
import zlib

import numpy as np

__all__ = ["crc32c", "blosc_decode", "blosc_encode", "zstd_decompress", "zstd_compress",
           "lz4_decompress", "have_zstd", "blosc_backend", "CodecUnavailable"]


class CodecUnavailable(RuntimeError):
    """No implementation of a block codec could be found on this host."""


# ------------------------------------------------------------------------------------ CRC-32C

def _crc32c_table() -> np.ndarray:
    poly = 0x82F63B78  # Castagnoli, reflected
    t = np.zeros(256, dtype=np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ poly if c & 1 else c >> 1
        t[i] = c
    return t


_CRC_TABLE = [int(v) for v in _crc32c_table()]


def crc32c(data) -> int:
    """CRC-32C (Castagnoli) of ``data`` -- the checksum the Zarr v3 ``crc32c`` codec appends.  Through liblsrecon's
    ``lsr_crc32c_host`` (SSE4.2 / slice-by-8, GIL released: a chunk-level checksum runs at memory speed from every
    reader thread); the byte loop below only where the library is not built (the shard index is a few hundred bytes)."""
    lib = _native_lib()
    if lib is not None and hasattr(lib, "lsr_crc32c_host"):
        buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data).view(np.uint8).reshape(-1)
        out = ctypes.c_uint32(0)
        rc = lib.lsr_crc32c_host(buf.ctypes.data if buf.size else None, buf.size, 0, ctypes.byref(out))
        if rc == 0:
            return int(out.value)
    return _crc32c_python(data)


def _crc32c_python(data) -> int:
    if len(data) > (4 << 20):
        import warnings

        warnings.warn(f"CRC-32C of {len(data) >> 20} MB in the pure-Python loop (~10 MB/s, holds the GIL): build "
                      "liblsrecon.so for the native checksum", RuntimeWarning, stacklevel=3)
    crc = 0xFFFFFFFF
    table = _CRC_TABLE
    for b in bytes(data):
        crc = table[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


# ------------------------------------------------------------------------------------ zstd / lz4

def _ctypes_lib(env: str, *names: str):
    path = os.environ.get(env)
    cands = [path] if path else []
    for n in names:
        found = ctypes.util.find_library(n)
        if found:
            cands.append(found)
        cands.append(f"lib{n}.so.1")
        cands.append(f"lib{n}.so")
    for c in cands:
        try:
            return ctypes.CDLL(c)
        except OSError:
            continue
    return None


class _Zstd:
    """Lazily picked zstd provider: (compress(bytes, level) -> bytes, decompress(bytes, n) -> bytes)."""

    def __init__(self):
        self.name = None
        self._c = self._d = None

    def _pick(self):
        if self.name is not None:
            return
        try:
            from numcodecs import Zstd as _NZ

            self._c = lambda b, level: bytes(_NZ(level=level).encode(b))
            self._d = lambda b, n: bytes(_NZ().decode(b))
            self.name = "numcodecs"
            return
        except ImportError:
            pass
        try:
            import zstandard as _z

            self._c = lambda b, level: _z.ZstdCompressor(level=level).compress(bytes(b))
            self._d = lambda b, n: _z.ZstdDecompressor().decompress(bytes(b), max_output_size=max(int(n), 1))
            self.name = "zstandard"
            return
        except ImportError:
            pass
        try:
            import pyarrow as pa

            if pa.Codec.is_available("zstd"):
                self._c = lambda b, level: pa.Codec("zstd", compression_level=level).compress(
                    bytes(b), asbytes=True)
                self._d = lambda b, n: pa.Codec("zstd").decompress(bytes(b), decompressed_size=int(n),
                                                                   asbytes=True)
                self.name = "pyarrow"
                return
        except ImportError:
            pass
        lib = _ctypes_lib("LSR_LIBZSTD", "zstd")
        if lib is not None:
            lib.ZSTD_compressBound.restype = ctypes.c_size_t
            lib.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
            lib.ZSTD_compress.restype = ctypes.c_size_t
            lib.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t,
                                          ctypes.c_int]
            lib.ZSTD_decompress.restype = ctypes.c_size_t
            lib.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
            lib.ZSTD_isError.restype = ctypes.c_uint
            lib.ZSTD_isError.argtypes = [ctypes.c_size_t]

            def comp(b, level):
                b = bytes(b)
                cap = lib.ZSTD_compressBound(len(b))
                buf = ctypes.create_string_buffer(cap)
                n = lib.ZSTD_compress(buf, cap, b, len(b), int(level))
                if lib.ZSTD_isError(n):
                    raise RuntimeError("ZSTD_compress failed")
                return buf.raw[:n]

            def decomp(b, n):
                b = bytes(b)
                buf = ctypes.create_string_buffer(max(int(n), 1))
                got = lib.ZSTD_decompress(buf, int(n), b, len(b))
                if lib.ZSTD_isError(got):
                    raise RuntimeError("ZSTD_decompress failed (corrupt block?)")
                return buf.raw[:got]

            self._c, self._d, self.name = comp, decomp, "libzstd"
            return
        raise CodecUnavailable(
            "zstd: none of numcodecs, zstandard, pyarrow or a system libzstd is available; install one "
            "(`pip install numcodecs`) or point LSR_LIBZSTD at libzstd.so")


_zstd = _Zstd()


def have_zstd() -> bool:
    try:
        _zstd._pick()
        return True
    except CodecUnavailable:
        return False


def zstd_compress(data, level: int = 1) -> bytes:
    _zstd._pick()
    return _zstd._c(data, level)


def zstd_decompress(data, nbytes: int) -> bytes:
    """``nbytes`` = the decompressed size (blosc stores it; for bare zstd chunks the caller knows
    the chunk shape)."""
    _zstd._pick()
    return _zstd._d(data, nbytes)


_zstd_direct = None
_zstd_direct_tried = False


def _zstd_lib():
    """The system libzstd through ctypes, for decoding straight into a caller's buffer (the Python
    providers above return a fresh ``bytes`` per stream, which then has to be copied)."""
    global _zstd_direct, _zstd_direct_tried
    if not _zstd_direct_tried:
        _zstd_direct_tried = True
        lib = _ctypes_lib("LSR_LIBZSTD", "zstd")
        if lib is not None:
            try:
                lib.ZSTD_decompress.restype = ctypes.c_size_t
                lib.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
                lib.ZSTD_isError.restype = ctypes.c_uint
                lib.ZSTD_isError.argtypes = [ctypes.c_size_t]
                _zstd_direct = lib
            except AttributeError:
                _zstd_direct = None
    return _zstd_direct


def zstd_decompress_into(src: np.ndarray, dst: np.ndarray) -> None:
    """Decode the zstd frame in ``src`` (uint8 view) into ``dst`` (writable contiguous uint8 view of
    exactly the decoded size)."""
    lib = _zstd_lib()
    if lib is not None:
        got = lib.ZSTD_decompress(dst.ctypes.data, dst.size, src.ctypes.data, src.size)   # releases the GIL
        if lib.ZSTD_isError(got) or got != dst.size:
            raise ValueError("corrupt blosc frame: zstd stream does not decode to the block size")
        return
    raw = zstd_decompress(src.tobytes(), dst.size)
    if len(raw) != dst.size:
        raise ValueError("corrupt blosc frame: zstd stream does not decode to the block size")
    dst[:] = np.frombuffer(raw, dtype=np.uint8)


def lz4_decompress(data, nbytes: int) -> bytes:
    """A raw LZ4 block (what blosc's lz4 / lz4hc compressors emit)."""
    try:
        import lz4.block as _l

        return _l.decompress(bytes(data), uncompressed_size=int(nbytes))
    except ImportError:
        pass
    try:
        import pyarrow as pa

        if pa.Codec.is_available("lz4_raw"):
            return pa.Codec("lz4_raw").decompress(bytes(data), decompressed_size=int(nbytes), asbytes=True)
    except ImportError:
        pass
    lib = _ctypes_lib("LSR_LIBLZ4", "lz4")
    if lib is None:
        raise CodecUnavailable("lz4: none of lz4, pyarrow or a system liblz4 is available")
    lib.LZ4_decompress_safe.restype = ctypes.c_int
    lib.LZ4_decompress_safe.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    buf = ctypes.create_string_buffer(max(int(nbytes), 1))
    got = lib.LZ4_decompress_safe(bytes(data), buf, len(data), int(nbytes))
    if got < 0:
        raise RuntimeError("LZ4_decompress_safe failed (corrupt block?)")
    return buf.raw[:got]


# ------------------------------------------------------------------------------------ blosc 1.x

BLOSC_VERSION_FORMAT = 2
_BLOSC_HEADER = 16
_MAX_SPLITS, _MIN_BUFFERSIZE = 16, 128           # c-blosc 1.x blosc.h
_F_SHUFFLE, _F_MEMCPYED, _F_BITSHUFFLE, _F_DONTSPLIT = 0x1, 0x2, 0x4, 0x10
_COMPRESSORS = {0: "blosclz", 1: "lz4", 2: "snappy", 3: "zlib", 4: "zstd"}
_COMP_CODE = {"lz4": 1, "zlib": 3, "zstd": 4}
_COMP_VERSIONLZ = {"lz4": 1, "zlib": 1, "zstd": 1}
SHUFFLE_NONE, SHUFFLE_BYTE, SHUFFLE_BIT = 0, 1, 2

_libblosc = None
_libblosc_tried = False


def _blosc_lib():
    global _libblosc, _libblosc_tried
    if not _libblosc_tried:
        _libblosc_tried = True
        if os.environ.get("LSR_BLOSC", "auto") != "python":
            lib = _ctypes_lib("LSR_LIBBLOSC", "blosc")
            if lib is not None:
                try:
                    lib.blosc_decompress_ctx.restype = ctypes.c_int
                    lib.blosc_decompress_ctx.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                                         ctypes.c_int]
                    lib.blosc_compress_ctx.restype = ctypes.c_int
                    lib.blosc_compress_ctx.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_size_t,
                                                       ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p,
                                                       ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t,
                                                       ctypes.c_int]
                    _libblosc = lib
                except AttributeError:
                    _libblosc = None
    return _libblosc


_numcodecs_blosc = None
_numcodecs_tried = False


def _numcodecs():
    """numcodecs' own blosc binding, when that package is installed (it is wherever iohub / zarr are)."""
    global _numcodecs_blosc, _numcodecs_tried
    if not _numcodecs_tried:
        _numcodecs_tried = True
        if os.environ.get("LSR_BLOSC", "auto") != "python":
            try:
                from numcodecs import blosc as nb

                _numcodecs_blosc = nb
            except ImportError:
                _numcodecs_blosc = None
    return _numcodecs_blosc


_native = None
_native_tried = False


def _native_lib():
    """This package's own frame walker (``lsr_blosc_decode_host`` in liblsrecon, ``csrc/blosc_frame.hip``:
    host code, the system libzstd / liblz4 / libz behind it) -- ``None`` when the library is not built."""
    global _native, _native_tried
    if not _native_tried:
        _native_tried = True
        if os.environ.get("LSR_BLOSC", "auto") not in ("python",):
            try:
                from .. import _lib

                lib = _lib.load()
                if hasattr(lib, "lsr_blosc_decode_host"):
                    _native = lib
            except Exception:  # noqa: BLE001 -- not built / not loadable here: the Python codec still reads the store
                _native = None
    return _native


def _native_decode(frame, dest: np.ndarray) -> bool:
    """Decode through ``lsr_blosc_decode_host``; ``False`` when the native walker does not take the
    frame (not built, bit shuffle, a compressor whose library is missing)."""
    lib = _native_lib()
    if lib is None:
        return False
    src = np.frombuffer(frame, dtype=np.uint8)
    rc = lib.lsr_blosc_decode_host(src.ctypes.data, src.size, dest.ctypes.data, dest.size, None)
    if rc == -3:                                     # LSR_E_UNSUPPORTED
        return False
    if rc != 0:
        raise ValueError(lib.lsr_last_error().decode("utf-8", "replace"))
    return True


def _native_encode(arr: np.ndarray, typesize: int, clevel: int, shuffle: int, blocksize: int):
    """One frame through ``lsr_blosc_encode_host`` (zstd streams, byte shuffle or none: the frames of
    :func:`_py_blosc_encode`, byte for byte, with the GIL released); ``None`` when the library is not built or
    libzstd's compressor is not loadable."""
    lib = _native_lib()
    if lib is None or not hasattr(lib, "lsr_blosc_encode_host") or not lib.lsr_blosc_host_encoder():
        return None
    lib.lsr_blosc_encode_bound.restype = ctypes.c_int64
    cap = int(lib.lsr_blosc_encode_bound(arr.size, typesize, blocksize))
    if cap < 0:
        return None
    buf = np.empty(cap, dtype=np.uint8)
    n = ctypes.c_int64(0)
    rc = lib.lsr_blosc_encode_host(arr.ctypes.data, arr.size, typesize, clevel, 1 if shuffle == SHUFFLE_BYTE else 0,
                                   blocksize, buf.ctypes.data, cap, ctypes.byref(n))
    if rc == -3:                                     # LSR_E_UNSUPPORTED
        return None
    if rc != 0:
        raise ValueError(lib.lsr_last_error().decode("utf-8", "replace"))
    return buf[:n.value].tobytes()


def blosc_backend() -> str:
    """Who decodes blosc frames: ``"libblosc"`` (ctypes) or ``"numcodecs"`` when a C blosc is
    loadable, ``"lsrecon"`` for this package's own frame walker over the system zstd, else
    ``"python"``."""
    if _blosc_lib() is not None:
        return "libblosc"
    if _numcodecs() is not None:
        return "numcodecs"
    return "lsrecon" if _native_lib() is not None else "python"


def blosc_header(frame) -> dict:
    if len(frame) < _BLOSC_HEADER:
        raise ValueError("blosc frame shorter than its 16-byte header")
    version, versionlz, flags, typesize, nbytes, blocksize, cbytes = struct.unpack_from("<BBBBIII", frame, 0)
    return dict(version=version, versionlz=versionlz, flags=flags, typesize=typesize, nbytes=nbytes,
                blocksize=blocksize, cbytes=cbytes, compressor=_COMPRESSORS.get(flags >> 5, "?"))


def _unshuffle(block: np.ndarray, typesize: int) -> np.ndarray:
    n = block.size // typesize
    out = np.empty_like(block)
    out[:n * typesize].reshape(n, typesize)[...] = block[:n * typesize].reshape(typesize, n).T
    out[n * typesize:] = block[n * typesize:]
    return out


def _unshuffle_into(block: np.ndarray, typesize: int, out: np.ndarray) -> None:
    """``out[...] = unshuffle(block)`` (both uint8, same size).  Two- and four-byte elements -- camera
    counts, float32 -- are assembled with whole-array shifts and ORs on the element type (3-4x the
    rate of the strided byte transpose on 256 KB blocks); other sizes take the transpose."""
    n = block.size // typesize
    wide = {2: np.uint16, 4: np.uint32}.get(typesize)
    body = out[:n * typesize]
    if wide is not None and n and body.ctypes.data % typesize == 0 and sys.byteorder == "little":
        dst = body.view(wide)
        np.copyto(dst, block[:n], casting="unsafe")
        tmp = np.empty(n, dtype=wide)
        for b in range(1, typesize):
            np.left_shift(block[b * n:(b + 1) * n], 8 * b, out=tmp, dtype=wide, casting="unsafe")
            np.bitwise_or(dst, tmp, out=dst)
    elif n:
        body.reshape(n, typesize)[...] = block[:n * typesize].reshape(typesize, n).T
    out[n * typesize:] = block[n * typesize:]


def _shuffle(block: np.ndarray, typesize: int) -> np.ndarray:
    n = block.size // typesize
    out = np.empty_like(block)
    out[:n * typesize].reshape(typesize, n)[...] = block[:n * typesize].reshape(n, typesize).T
    out[n * typesize:] = block[n * typesize:]
    return out


def _bitshuffle_elems(nbytes: int, typesize: int) -> int:
    """c-blosc 1.x bit-shuffles a block only when its element count is a multiple of 8 and leaves it
    verbatim otherwise (``blosc_internal_bitshuffle``; checked against libblosc 1.21.0)."""
    n = nbytes // typesize
    return n if n % 8 == 0 else 0


def _bitunshuffle(block: np.ndarray, typesize: int) -> np.ndarray:
    """Inverse of the bitshuffle transform (bytes past the last whole element are verbatim)."""
    n = _bitshuffle_elems(block.size, typesize)
    out = block.copy()
    if n:
        rows = block[:n * typesize].reshape(typesize * 8, n // 8)
        bits = np.unpackbits(rows, axis=1, bitorder="little")            # (typesize*8, n)
        out[:n * typesize] = np.packbits(bits.T, axis=1, bitorder="little").reshape(-1)
    return out


def _bitshuffle(block: np.ndarray, typesize: int) -> np.ndarray:
    n = _bitshuffle_elems(block.size, typesize)
    out = block.copy()
    if n:
        bits = np.unpackbits(block[:n * typesize].reshape(n, typesize), axis=1, bitorder="little")
        out[:n * typesize] = np.packbits(bits.T, axis=1, bitorder="little").reshape(-1)
    return out


def _decode_stream(comp: str, data, nbytes: int) -> bytes:
    if comp == "zstd":
        return zstd_decompress(data, nbytes)
    if comp == "lz4":
        return lz4_decompress(data, nbytes)
    if comp == "zlib":
        return zlib.decompress(bytes(data))
    raise CodecUnavailable(f"blosc compressor {comp!r} is not implemented here (zstd, lz4, zlib are); "
                           "install numcodecs or a system libblosc")


def _stream_into(comp: str, src: np.ndarray, dst: np.ndarray) -> None:
    """One blosc stream (``src``: its cbytes) into ``dst`` (its decoded bytes).  Whatever a decoder raises on a
    damaged stream (``zlib.error``, a provider's own exception type) leaves here as ``ValueError``."""
    if src.size == dst.size:                        # stored
        dst[:] = src
        return
    try:
        if comp == "zstd":
            zstd_decompress_into(src, dst)
            return
        raw = _decode_stream(comp, memoryview(src), dst.size)
    except (ValueError, CodecUnavailable, MemoryError):
        raise
    except Exception as exc:  # noqa: BLE001 -- decoder-specific error types of four possible providers
        raise ValueError(f"corrupt blosc frame: the {comp} stream does not decode ({type(exc).__name__}: {exc})") from exc
    if len(raw) != dst.size:
        raise ValueError("corrupt blosc frame: block size mismatch")
    dst[:] = np.frombuffer(raw, dtype=np.uint8)


def _py_blosc_decode(frame, out: np.ndarray) -> None:
    """Decode a frame into ``out`` (uint8, the frame's ``nbytes``): the Python statement of the walk
    that ``lsr_blosc_decode_host`` does natively -- last resort and cross-check."""
    h = blosc_header(frame)
    nbytes, blocksize, typesize, flags = h["nbytes"], h["blocksize"], h["typesize"], h["flags"]
    if out.size != nbytes:
        raise ValueError(f"blosc frame holds {nbytes} bytes, destination has {out.size}")
    src = np.frombuffer(frame, dtype=np.uint8)
    if flags & _F_MEMCPYED:
        if src.size < _BLOSC_HEADER + nbytes:
            raise ValueError("corrupt blosc frame: stream runs past the end")
        out[:] = src[_BLOSC_HEADER:_BLOSC_HEADER + nbytes]
        return
    if nbytes == 0:
        return
    typesize = typesize or 1            # (a zero in the header: one-byte elements, as the native walker reads it)
    byte_shuffled = bool(flags & _F_SHUFFLE) and typesize > 1
    bit_shuffled = not byte_shuffled and bool(flags & _F_BITSHUFFLE)
    if not 0 < blocksize <= nbytes:
        raise ValueError(f"corrupt blosc frame: blocksize {blocksize} of {nbytes} bytes")
    if (not (flags & _F_DONTSPLIT) and typesize <= _MAX_SPLITS and blocksize // typesize >= _MIN_BUFFERSIZE
            and blocksize % typesize):
        raise ValueError(f"corrupt blosc frame: split blocks of {blocksize} bytes are not a multiple of typesize {typesize}")
    nblocks = -(-nbytes // blocksize)
    if _BLOSC_HEADER + 4 * nblocks > src.size:
        raise ValueError("corrupt blosc frame: block table runs past the end")
    bstarts = struct.unpack_from(f"<{nblocks}i", frame, _BLOSC_HEADER)
    comp = h["compressor"]
    scratch = np.empty(blocksize, dtype=np.uint8) if byte_shuffled or bit_shuffled else None
    for b in range(nblocks):
        bsize = min(blocksize, nbytes - b * blocksize)
        leftover = bsize != blocksize
        split = (not (flags & _F_DONTSPLIT) and typesize <= _MAX_SPLITS
                 and blocksize // typesize >= _MIN_BUFFERSIZE and not leftover)
        nsplits = typesize if split else 1
        neblock = bsize // nsplits
        dest = out[b * blocksize:b * blocksize + bsize]
        target = dest if scratch is None else scratch[:bsize]
        pos = bstarts[b]
        for k in range(nsplits):
            if pos < 0 or pos + 4 > src.size:
                raise ValueError("corrupt blosc frame: stream runs past the end")
            (cb,) = struct.unpack_from("<i", frame, pos)
            pos += 4
            if cb < 0 or pos + cb > src.size:
                raise ValueError("corrupt blosc frame: stream runs past the end")
            _stream_into(comp, src[pos:pos + cb], target[k * neblock:(k + 1) * neblock])
            pos += cb
        if byte_shuffled:
            _unshuffle_into(target, typesize, dest)
        elif bit_shuffled:
            dest[:] = _bitunshuffle(target, typesize)


def blosc_decode(frame, out=None, backend: str | None = None) -> np.ndarray:
    """Decode one blosc 1.x frame.  ``out`` (a writable, C-contiguous array of the right byte size, e.g.
    a slab of a pinned staging buffer) receives the bytes in place; returns it as uint8 otherwise."""
    h = blosc_header(frame)
    if out is None:
        dest = np.empty(h["nbytes"], dtype=np.uint8)
    else:
        if not out.flags.c_contiguous or not out.flags.writeable:
            raise ValueError("blosc_decode: out must be a writable C-contiguous array")
        dest = out.reshape(-1).view(np.uint8)
        if dest.size != h["nbytes"]:
            raise ValueError(f"blosc frame holds {h['nbytes']} bytes, destination has {dest.size}")
    lib = _blosc_lib() if backend in (None, "libblosc") else None
    if backend == "libblosc" and lib is None:
        raise CodecUnavailable("libblosc requested but not loadable")
    if lib is not None:
        buf = bytes(frame) if not isinstance(frame, bytes) else frame
        got = lib.blosc_decompress_ctx(buf, dest.ctypes.data, dest.size, 1)
        if got != h["nbytes"]:
            raise ValueError(f"blosc_decompress_ctx returned {got}, expected {h['nbytes']}")
    elif backend is None and _numcodecs() is not None:
        _numcodecs().decompress(bytes(frame), dest)
    elif backend in (None, "lsrecon") and _native_decode(frame, dest):
        pass
    elif backend == "lsrecon":
        raise CodecUnavailable("the native frame walker (liblsrecon) is not built or does not take this frame")
    else:
        _py_blosc_decode(frame, dest)
    return dest if out is None else out


def _py_blosc_encode(data: np.ndarray, typesize: int, cname: str, clevel: int, shuffle: int,
                     blocksize: int) -> bytes:
    if cname not in ("zstd", "zlib"):
        raise CodecUnavailable(f"pure-Python blosc encoder writes zstd or zlib streams, not {cname!r}")
    nbytes = data.size
    if not blocksize:
        blocksize = 256 * 1024
    blocksize = max(typesize, min(blocksize, nbytes) // typesize * typesize) if nbytes >= typesize else max(nbytes, 1)
    flags = _F_DONTSPLIT | (_COMP_CODE[cname] << 5)   # one stream per block: the simplest valid frame
    if shuffle == SHUFFLE_BYTE and typesize > 1:
        flags |= _F_SHUFFLE
    elif shuffle == SHUFFLE_BIT:
        flags |= _F_BITSHUFFLE
    nblocks = -(-nbytes // blocksize) if nbytes else 0
    streams, pos = [], _BLOSC_HEADER + 4 * nblocks
    bstarts = []
    for b in range(nblocks):
        block = data[b * blocksize:(b + 1) * blocksize]
        if flags & _F_SHUFFLE:
            block = _shuffle(block, typesize)
        elif flags & _F_BITSHUFFLE:
            block = _bitshuffle(block, typesize)
        raw = block.tobytes()
        comp = zstd_compress(raw, clevel) if cname == "zstd" else zlib.compress(raw, clevel)
        if len(comp) >= len(raw):
            comp = raw                                 # stored: cbytes == neblock
        bstarts.append(pos)
        streams.append(struct.pack("<i", len(comp)) + comp)
        pos += 4 + len(comp)
    if pos >= nbytes + _BLOSC_HEADER:                  # incompressible: the memcpyed form
        flags = (flags | _F_MEMCPYED)
        body = data.tobytes()
        return struct.pack("<BBBBIII", BLOSC_VERSION_FORMAT, _COMP_VERSIONLZ[cname], flags, typesize, nbytes,
                           blocksize, _BLOSC_HEADER + nbytes) + body
    head = struct.pack("<BBBBIII", BLOSC_VERSION_FORMAT, _COMP_VERSIONLZ[cname], flags, typesize, nbytes,
                       blocksize, pos)
    return head + struct.pack(f"<{nblocks}i", *bstarts) + b"".join(streams)


def blosc_encode(data, typesize: int, cname: str = "zstd", clevel: int = 1, shuffle: int = SHUFFLE_BYTE,
                 blocksize: int = 0, backend: str | None = None) -> bytes:
    """One blosc 1.x frame of ``data`` (any C-contiguous array; at most 2 GiB - 16, the format's
    limit).  Defaults are the acquisition's: zstd, level 1, byte shuffle."""
    arr = np.ascontiguousarray(data).reshape(-1).view(np.uint8)
    if arr.size > 0x7FFFFFFF - _BLOSC_HEADER:
        raise ValueError("a blosc 1.x frame holds less than 2 GiB; use smaller chunks")
    typesize = int(typesize)
    if not 1 <= typesize <= 255:
        typesize = 1
    lib = _blosc_lib() if backend in (None, "libblosc") else None
    if backend == "libblosc" and lib is None:
        raise CodecUnavailable("libblosc requested but not loadable")
    if lib is not None:
        cap = arr.size + _BLOSC_HEADER
        buf = ctypes.create_string_buffer(cap)
        n = lib.blosc_compress_ctx(int(clevel), int(shuffle), typesize, arr.size, arr.ctypes.data, buf, cap,
                                   cname.encode(), int(blocksize), 1)
        if n <= 0:
            raise RuntimeError(f"blosc_compress_ctx failed ({n})")
        return buf.raw[:n]
    if backend in (None, "lsrecon") and cname == "zstd" and int(shuffle) in (SHUFFLE_NONE, SHUFFLE_BYTE):
        frame = _native_encode(arr, typesize, int(clevel), int(shuffle), int(blocksize))
        if frame is not None:
            return frame
    if backend == "lsrecon":
        raise CodecUnavailable("the native frame encoder (liblsrecon) is not built or does not take these parameters")
    return _py_blosc_encode(arr, typesize, cname, int(clevel), int(shuffle), int(blocksize))
