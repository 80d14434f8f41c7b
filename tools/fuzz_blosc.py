"""Mutated blosc frames through the native frame walker (``lsr_blosc_decode_host``) and the Python walker.

    python tools/fuzz_blosc.py [--seconds 60] [--seed 1]        # best under tools/host_sanitize.sh's library:
    LSR_LIBRARY=/tmp/lsr_asan/liblsrecon.so LD_PRELOAD=<libclang_rt.asan-x86_64.so> python tools/fuzz_blosc.py

Valid frames (zstd / zlib; byte shuffle, bit shuffle, none; typesize 1-8; blocks split and not)
are damaged -- header fields overwritten, block offsets redirected, bytes flipped, the frame truncated or padded --
and decoded into an exactly sized buffer.  Every outcome must be an error status / ValueError or a completed decode;
never a crash, a hang or (under ASan) an out-of-bounds access.  Prints one JSON line.
"""

from __future__ import annotations

import argparse
import ctypes
import json
import sys
import time

from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()

    from shrimpy_amd import _lib
    from shrimpy_amd.io import codecs

    lib = _lib.load()
    rng = np.random.default_rng(args.seed)
    # (the frames are made by this package's encoder, which writes zstd and zlib streams)
    cnames = [c for c, code in (("zstd", 4), ("zlib", 3)) if lib.lsr_blosc_host_codec(code)]
    stats = {"frames": 0, "native_ok": 0, "native_refused": 0, "python_ok": 0, "python_refused": 0, "codecs": cnames}
    t_end = time.perf_counter() + args.seconds
    while time.perf_counter() < t_end:
        typesize = int(rng.choice([1, 2, 4, 8]))
        n = int(rng.integers(1, 40000)) * typesize
        kind = rng.integers(0, 3)
        data = (rng.integers(0, 4, n, dtype=np.uint8) if kind == 0 else
                rng.integers(0, 256, n, dtype=np.uint8) if kind == 1 else np.zeros(n, np.uint8))
        frame = bytearray(codecs.blosc_encode(data, typesize, cname=str(rng.choice(cnames)), clevel=int(rng.integers(1, 6)),
                                              shuffle=int(rng.choice([0, 1, 2])),
                                              blocksize=int(rng.choice([0, 256, 4096, 32768]))))
        for _ in range(int(rng.integers(1, 6))):
            how = rng.integers(0, 6)
            if how == 0 and len(frame) > 16:          # a header field
                pos = int(rng.integers(0, 16))
                frame[pos] = int(rng.integers(0, 256))
            elif how == 1 and len(frame) > 20:        # a block offset
                pos = 16 + 4 * int(rng.integers(0, max(1, (len(frame) - 16) // 4 // 8 + 1)))
                if pos + 4 <= len(frame):
                    frame[pos:pos + 4] = int(rng.integers(0, 2**32)).to_bytes(4, "little")
            elif how == 2:                            # a flipped byte anywhere
                pos = int(rng.integers(0, len(frame)))
                frame[pos] ^= 1 << int(rng.integers(0, 8))
            elif how == 3 and len(frame) > 1:         # truncation
                del frame[int(rng.integers(1, len(frame))):]
            elif how == 4:                            # trailing garbage
                frame += bytes(rng.integers(0, 256, int(rng.integers(1, 64)), dtype=np.uint8))
            else:                                     # a wrong size claim
                frame[4:8] = int(rng.integers(0, 2**31)).to_bytes(4, "little")
        stats["frames"] += 1
        src = np.frombuffer(bytes(frame), dtype=np.uint8)
        out = np.empty(n, np.uint8)
        rc = lib.lsr_blosc_decode_host(src.ctypes.data, src.size, out.ctypes.data, out.size, None)
        stats["native_ok" if rc == 0 else "native_refused"] += 1
        try:
            if len(frame) >= 16 and int.from_bytes(frame[4:8], "little") > (64 << 20):
                raise ValueError("size claim beyond this fuzzer's budget")      # (np.empty of 2 GiB per frame)
            got = codecs.blosc_decode(bytes(frame), backend="python")
            stats["python_ok"] += 1
            if rc == 0 and got.size == out.size and not np.array_equal(got, out):
                stats["both_ok_but_different"] = stats.get("both_ok_but_different", 0) + 1
        except (ValueError, codecs.CodecUnavailable):
            stats["python_refused"] += 1
    print(json.dumps(stats))


if __name__ == "__main__":
    main()
