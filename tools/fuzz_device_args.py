"""Random arguments through every DEVICE entry point of the C ABI on a machine WITHOUT a GPU: whatever the host side does
with them before a launch -- shape arithmetic, table sizes, tile plans, LDS budgets -- must end in a status code, not in a
crash or (against a library whose host code is built with ASan + UBSan, see tools/host_sanitize.sh [scratch-dir] [fuzz-seconds]) a report:
no signed overflow, no division by zero, no read past a host table.  Nothing can launch (there is no device; a call that
passes validation fails in the HIP runtime with a status), so the pointers are host buffers nobody dereferences except
the documented host-pointer arguments (taps, matrices, shape outputs), which are sized for the documented maxima.

    LSR_LIBRARY=<sanitized liblsrecon.so> LD_PRELOAD=<libclang_rt.asan-x86_64.so> python tools/fuzz_device_args.py
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

# (host code that reads its buffer arguments -- the frame walker, the CRC -- has its own fuzzers: tools/fuzz_blosc.py,
# tests/test_zarr_codecs.py)
SKIP = {"lsr_blosc_decode_host", "lsr_pinned_free", "lsr_pinned_alloc", "lsr_set_host_threads", "lsr_get_host_threads",
        "lsr_version", "lsr_crc32c_host", "lsr_crc32c_host_portable", "lsr_source_sha16", "lsr_blosc_encode_host",
        "lsr_blosc_encode_bound", "lsr_blosc_host_encoder"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()

    import torch

    if torch.cuda.is_available():
        raise SystemExit("this fuzzer is for a machine without a GPU: with one, a call that passes validation would launch")
    from shrimpy_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(args.seed)
    buf = np.ascontiguousarray(rng.random(1 << 20))          # 8 MB of finite doubles / floats / whatever is read
    ints = [-(1 << 40), -1, 0, 1, 2, 3, 4, 5, 7, 8, 9, 11, 15, 16, 17, 63, 64, 65, 100, 256, 1000, 2048, 4096, 65535, 65536,
            (1 << 30) - 1, 1 << 30, (1 << 31) - 1, 1 << 31, 1 << 32, 1 << 40, (1 << 62)]
    small = [-1, 0, 1, 2, 3, 4, 5, 7, 9, 15, 16, 17, 256, 257, (1 << 31) - 1]
    floats = [0.0, -0.0, 1.0, -1.0, 0.5, 1e-6, 1e30, float("inf"), float("nan")]
    names = sorted(n for n in _lib.SIGNATURES if n not in SKIP and not n.endswith("_cpu"))
    stats = {"entries": len(names), "calls": 0, "status_0": 0, "status_negative": 0, "status_positive": 0}
    t_end = time.perf_counter() + args.seconds
    while time.perf_counter() < t_end:
        name = names[int(rng.integers(0, len(names)))]
        sig = _lib.SIGNATURES[name]
        call = []
        for k, t in enumerate(sig):
            last = k == len(sig) - 1
            if t is ctypes.c_int64:
                call.append(int(rng.choice(ints)) if rng.random() < 0.6 else int(rng.integers(1, 70)))
            elif t is ctypes.c_int:
                call.append(int(rng.choice(small)) if rng.random() < 0.6 else int(rng.integers(0, 12)))
            elif t is ctypes.c_float:
                call.append(ctypes.c_float(float(rng.choice(floats))))
            elif t is ctypes.c_double:
                call.append(ctypes.c_double(float(rng.choice(floats))))
            elif t is ctypes.c_uint32:
                call.append(int(rng.integers(0, 1 << 32)))
            elif t is ctypes.c_void_p and last:
                call.append(None)                                  # the stream
            elif t is ctypes.c_void_p:
                call.append(None if rng.random() < 0.05 else buf.ctypes.data + 16 * int(rng.integers(0, 8)))
            else:                                                  # typed pointers: matrices, shape / index outputs
                call.append(ctypes.cast(buf.ctypes.data + 8 * int(rng.integers(0, 64)), t))
        rc = int(getattr(lib, name)(*call))
        stats["calls"] += 1
        stats["status_0" if rc == 0 else "status_negative" if rc < 0 else "status_positive"] += 1
    print(json.dumps(stats))


if __name__ == "__main__":
    main()
