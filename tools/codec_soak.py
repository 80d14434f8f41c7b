"""Random volumes through the DEVICE codecs, compared with their host twins and with libzstd (run on the GPU box).

    python tools/codec_soak.py [--seconds 120] [--seed 1] > profiles/rNN_codec_soak.json

Per round: a volume of 1 / 2 / 4-byte elements (camera counts, smooth results, constant, incompressible, runs, sparse),
a chunk size and a block size;
* device encoder: frames byte for byte those of the encoder twin; decoded by libzstd (the Python frame walker) to the
  volume;
* device decoder on those frames and on frames libzstd wrote itself (levels 1-9): the volume;
* the same libzstd frames with damaged bytes: the status word names a block or the decode completes; the process and
  the card survive (every access of the kernel is bounded by the frame and the block -- the twin runs the same source
  under ASan in tools/host_sanitize.sh).
One JSON line.  Sizes stay below 8 MB per volume so that the host side (numpy, libzstd) keeps up.
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
sys.path.insert(0, str(ROOT / "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()

    import torch

    from fuzz_device_codec import contents
    from shrimpy_amd import _lib
    from shrimpy_amd.io import codecs
    from shrimpy_amd.io.device_codec import (DecodeError, DeviceBloscDecoder, DeviceBloscEncoder, encode_frames_host,
                                             frame_layout)

    dev = torch.device("cuda:0")
    rng = np.random.default_rng(args.seed)
    stats = {"volumes": 0, "bytes": 0, "device_frames_equal_twin": 0, "libzstd_frames_decoded": 0, "damaged": 0,
             "damaged_refused": 0, "damaged_decoded": 0, "library": _lib.kernel_source_sha16()}
    t_end = time.perf_counter() + args.seconds
    t_note = time.perf_counter()
    while time.perf_counter() < t_end:
        dtype = [np.uint8, np.uint16, np.float32][int(rng.integers(0, 3))]
        item = np.dtype(dtype).itemsize
        n = int(rng.choice([1, 9, 5000, 70000, 1 << 20, int(rng.integers(1, 2_000_000))]))
        vol = contents(rng, int(rng.integers(0, 6)), n, dtype)
        raw = vol.view(np.uint8).reshape(-1)
        frame_bytes = int(rng.choice([4096, 65536, 1 << 20, vol.nbytes])) // item * item or item
        blocksize = int(rng.choice([0, 0, 4096, 32768, 65536 * item]))
        dvol = torch.from_numpy(vol).to(dev)
        # -- encoder
        enc = DeviceBloscEncoder(vol.nbytes, item, frame_bytes, dev, blocksize)
        frames = enc.encode_to_host(dvol)
        twin = encode_frames_host(vol, frame_bytes, blocksize)
        assert frames == twin, ("device encoder differs from its twin", dtype, n, frame_bytes, blocksize)
        stats["device_frames_equal_twin"] += len(frames)
        back = np.concatenate([codecs.blosc_decode(f, backend="python") for f in frames[:4]])
        assert np.array_equal(back[:min(back.size, raw.size)], raw[:min(back.size, raw.size)]), ("libzstd on device frames", dtype, n)
        # -- decoder on the encoder's frames
        lay = frame_layout(frames[0])
        dec = DeviceBloscDecoder(len(frames) * frame_bytes, frame_bytes, lay["blocksize"], item, dev)
        out = torch.empty(len(frames) * frame_bytes, dtype=torch.uint8, device=dev)
        dec.decode_from_host(frames, out)
        assert np.array_equal(out.cpu().numpy()[:raw.size], raw), ("device decoder on device frames", dtype, n)
        # -- decoder on libzstd's own frames, one chunk
        m = min(raw.size, int(rng.integers(1, 400000)) * item)
        piece = raw[:m]
        frame = codecs.blosc_encode(piece, item, cname="zstd", clevel=int(rng.integers(1, 10)), shuffle=1 if item > 1 else 0,
                                    blocksize=int(rng.choice([0, 4096, 32768, 65536])))
        lay = frame_layout(frame)
        if lay is not None:
            dec = DeviceBloscDecoder(m, lay["nbytes"], lay["blocksize"], lay["typesize"], dev)
            out = torch.empty(m, dtype=torch.uint8, device=dev)
            dec.decode_from_host([frame], out)
            assert np.array_equal(out.cpu().numpy(), piece), ("device decoder on a libzstd frame", dtype, m)
            stats["libzstd_frames_decoded"] += 1
            for _ in range(4):
                bad = bytearray(frame)
                for _ in range(int(rng.integers(1, 4))):
                    pos = int(rng.integers(16, len(bad)))
                    if rng.integers(0, 3) == 0 and pos + 4 <= len(bad):
                        bad[pos:pos + 4] = int(rng.integers(0, 2**31)).to_bytes(4, "little")
                    else:
                        bad[pos] ^= 1 << int(rng.integers(0, 8))
                if rng.integers(0, 4) == 0 and len(bad) > 40:
                    del bad[int(rng.integers(20, len(bad))):]
                stats["damaged"] += 1
                try:
                    dec.decode_from_host([bytes(bad)], out)
                    stats["damaged_decoded"] += 1
                except DecodeError:
                    stats["damaged_refused"] += 1
        stats["volumes"] += 1
        stats["bytes"] += int(raw.size)
        if time.perf_counter() - t_note > 30:
            print(f"[codec_soak] {stats['volumes']} volumes", file=sys.stderr, flush=True)
            t_note = time.perf_counter()
    torch.cuda.synchronize()
    print(json.dumps(stats))


if __name__ == "__main__":
    main()
