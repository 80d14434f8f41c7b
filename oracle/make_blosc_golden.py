"""Generate ``tests/golden/blosc_frames.npz``: frames produced by a REAL c-blosc (libblosc 1.21.0 from
the build image, through ctypes) for the pure-Python blosc decoder of ``shrimpy_amd/io/codecs.py`` to be
pinned against.  TEST INFRASTRUCTURE: run in the build container, commit the output.

    python oracle/make_blosc_golden.py [/path/to/libblosc.so]
"""

import ctypes
import sys

from pathlib import Path

import numpy as np

LIB = sys.argv[1] if len(sys.argv) > 1 else "/opt/conda/lib/libblosc.so.1"
OUT = Path(__file__).resolve().parent.parent / "tests" / "golden" / "blosc_frames.npz"


def main():
    b = ctypes.CDLL(LIB)
    b.blosc_get_version_string.restype = ctypes.c_char_p
    b.blosc_compress_ctx.restype = ctypes.c_int
    b.blosc_compress_ctx.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p,
                                     ctypes.c_size_t, ctypes.c_int]
    print("libblosc", b.blosc_get_version_string().decode())
    rng = np.random.default_rng(11)
    cases = {
        # name: (data, cname, shuffle, blocksize)
        "u16_zstd_shuffle_split": (rng.integers(80, 600, 9000).astype(np.uint16), b"zstd", 1, 4096),
        "u16_zstd_shuffle_auto": (rng.integers(80, 600, 40001).astype(np.uint16), b"zstd", 1, 0),
        "u16_zstd_noshuffle": (rng.integers(80, 600, 5000).astype(np.uint16), b"zstd", 0, 0),
        "u16_zstd_bitshuffle": (rng.integers(80, 600, 4096).astype(np.uint16), b"zstd", 2, 0),
        "u16_zstd_bitshuffle_ragged": (rng.integers(80, 600, 565).astype(np.uint16), b"zstd", 2, 0),
        "f32_lz4_shuffle_split": (rng.integers(0, 4000, 20000).astype(np.float32), b"lz4", 1, 8192),
        "f32_lz4hc_shuffle": (rng.integers(0, 4000, 3000).astype(np.float32), b"lz4hc", 1, 0),
        "f32_zlib_shuffle": (rng.integers(0, 4000, 3001).astype(np.float32), b"zlib", 1, 0),
        "f64_zstd_bitshuffle": (rng.integers(0, 9, 2048).astype(np.float64), b"zstd", 2, 0),
        "u8_zstd_incompressible": (rng.integers(0, 256, 4000).astype(np.uint8), b"zstd", 1, 0),
        "u16_zstd_zeros": (np.zeros(33000, np.uint16), b"zstd", 1, 0),
        "u16_zstd_tiny": (np.array([1, 2, 3], np.uint16), b"zstd", 1, 0),
        "u16_zstd_empty": (np.zeros(0, np.uint16), b"zstd", 1, 0),
    }
    out = {}
    for name, (a, cname, shuffle, bs) in cases.items():
        cap = a.nbytes + 16
        buf = ctypes.create_string_buffer(cap)
        n = b.blosc_compress_ctx(1, shuffle, a.itemsize, a.nbytes, a.ctypes.data, buf, cap, cname, bs, 1)
        assert n > 0, name
        out[name + ".frame"] = np.frombuffer(buf.raw[:n], np.uint8)
        out[name + ".data"] = a
        print(f"{name}: {a.nbytes} -> {n} bytes, flags {buf.raw[2]:#04x}")
    np.savez_compressed(OUT, **out)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
