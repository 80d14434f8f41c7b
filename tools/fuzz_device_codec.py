"""The device codecs' host twins under random and damaged input.

    python tools/fuzz_device_codec.py [--seconds 60] [--seed 1]        # best under tools/host_sanitize.sh's library

The twins (``lsr_blosc_decode_device_cpu`` / ``lsr_blosc_encode_device_cpu``) are the kernels' own source compiled for
the host (``csrc/zstd_lane.hpp``, ``csrc/zstd_huf.hpp``), so an out-of-range access found here under ASan is one the
kernel would make on the card, where nothing reports it.  Three legs per round:

* volumes of 1 / 2 / 4-byte elements with contents from constant to incompressible, encoded by the encoder twin, read
  back by libzstd through the Python frame walker AND by the decoder twin: both must return the volume;
* frames written by this package's host encoder (libzstd levels 1-19, blocks of 4-256 KB) through the decoder twin:
  the volume again;
* those frames damaged (flipped bytes, truncation, redirected block offsets, wrong size claims): an error status or a
  completed decode into the exactly sized buffer -- never a crash or an access outside the two buffers.

Prints one JSON line.
"""

from __future__ import annotations

import argparse
import json
import sys
import time

from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def contents(rng, kind: int, n: int, dtype) -> np.ndarray:
    """Volumes an acquisition and a reconstruction produce, and the corner cases around them."""
    if kind == 0:                                   # camera counts: offset + shot noise
        v = 100 + rng.poisson(rng.choice([2.0, 30.0, 900.0]), n)
    elif kind == 1:                                 # a smooth positive result
        t = np.linspace(0, rng.uniform(3, 90), n)
        v = (1 + np.sin(t)) * rng.uniform(1, 4000) + rng.normal(0, rng.choice([0.0, 0.5, 30.0]), n)
    elif kind == 2:                                 # constant
        v = np.full(n, rng.integers(0, 250))
    elif kind == 3:                                 # incompressible
        return rng.integers(0, 256, n * np.dtype(dtype).itemsize, dtype=np.uint8).view(dtype)[:n].copy()
    elif kind == 4:                                 # two values, long runs
        v = np.repeat(rng.integers(0, 2, n // 37 + 1) * 1000, 37)[:n]
    else:                                           # sparse
        v = np.zeros(n)
        v[rng.integers(0, n, max(1, n // 50))] = rng.integers(1, 60000)
    if np.dtype(dtype).kind == "f":
        return v.astype(dtype)
    return np.clip(v, 0, np.iinfo(dtype).max).astype(dtype)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()

    from shrimpy_amd.io import codecs
    from shrimpy_amd.io.device_codec import DecodeError, decode_frames_host, encode_frames_host, frame_layout

    rng = np.random.default_rng(args.seed)
    stats = {"encoded_volumes": 0, "libzstd_frames": 0, "damaged": 0, "damaged_refused": 0, "damaged_decoded": 0,
             "not_in_layout": 0}
    t_end = time.perf_counter() + args.seconds
    while time.perf_counter() < t_end:
        dtype = [np.uint8, np.uint16, np.float32][int(rng.integers(0, 3))]
        item = np.dtype(dtype).itemsize
        # -- leg 1: encoder twin -> libzstd and the decoder twin
        n = int(rng.choice([1, 7, 300, 4096, 70000, 300000, int(rng.integers(1, 600000))]))
        vol = contents(rng, int(rng.integers(0, 6)), n, dtype)
        frame_bytes = int(rng.choice([4096, 65536, 1 << 20, vol.nbytes + int(rng.integers(0, 3)) * item])) // item * item or item
        blocksize = int(rng.choice([0, 0, 4096, 32768, 65536 * item]))
        frames = encode_frames_host(vol, frame_bytes, blocksize)
        lay = frame_layout(frames[0])
        back = np.concatenate([codecs.blosc_decode(f, backend="python") for f in frames])[:vol.nbytes]
        assert np.array_equal(back, vol.view(np.uint8).reshape(-1)), ("libzstd disagrees with the encoder twin", dtype, n, frame_bytes)
        got = decode_frames_host(frames, frame_bytes, lay["blocksize"], item, len(frames) * frame_bytes)
        assert np.array_equal(got[:vol.nbytes], vol.view(np.uint8).reshape(-1)), ("decoder twin", dtype, n, frame_bytes)
        stats["encoded_volumes"] += 1
        # -- leg 2: libzstd's own frames -> decoder twin
        n = int(rng.integers(1, 200000))
        vol = contents(rng, int(rng.integers(0, 6)), n, dtype)
        raw = vol.view(np.uint8).reshape(-1)
        frame = codecs.blosc_encode(raw, item, cname="zstd", clevel=int(rng.integers(1, 10)), shuffle=1 if item > 1 else int(rng.integers(0, 2)),
                                    blocksize=int(rng.choice([0, 4096, 32768, 65536, 262144])))
        lay = frame_layout(frame)
        if lay is None:
            stats["not_in_layout"] += 1
            continue
        got = decode_frames_host([frame], lay["nbytes"], lay["blocksize"], lay["typesize"], raw.size)
        assert np.array_equal(got, raw), ("decoder twin on a libzstd frame", dtype, n)
        stats["libzstd_frames"] += 1
        # -- leg 3: the same frame, damaged
        for _ in range(8):
            bad = bytearray(frame)
            for _ in range(int(rng.integers(1, 5))):
                how = int(rng.integers(0, 5))
                if how == 0:
                    bad[int(rng.integers(16, len(bad)))] ^= 1 << int(rng.integers(0, 8))
                elif how == 1 and len(bad) > 40:
                    del bad[int(rng.integers(17, len(bad))):]
                elif how == 2 and len(bad) > 24:                   # a block offset
                    pos = 16 + 4 * int(rng.integers(0, max(1, -(-lay["nbytes"] // lay["blocksize"]))))
                    if pos + 4 <= len(bad):
                        bad[pos:pos + 4] = int(rng.integers(0, 2**32)).to_bytes(4, "little")
                elif how == 3:                                     # a compressed-size word inside a block
                    pos = int(rng.integers(16, max(17, len(bad) - 4)))
                    bad[pos:pos + 4] = int(rng.integers(0, 2**31)).to_bytes(4, "little")
                else:
                    bad[int(rng.integers(16, len(bad)))] = int(rng.integers(0, 256))
            stats["damaged"] += 1
            try:
                out = decode_frames_host([bytes(bad)], lay["nbytes"], lay["blocksize"], lay["typesize"], raw.size)
                assert out.size == raw.size
                stats["damaged_decoded"] += 1
            except DecodeError:
                stats["damaged_refused"] += 1
    print(json.dumps(stats))


if __name__ == "__main__":
    main()
