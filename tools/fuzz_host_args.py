"""Random, mostly invalid arguments through the host twins' C entry points: an error status or a completed call, never an
out-of-bounds access (run it against the sanitized library of tools/host_sanitize.sh).  Buffers are sized for the largest
shape the generator can ask for; dimensions, tap counts, strides, modes and epilogues are drawn from ranges that include
zero, negatives and values past the documented limits.  Prints one JSON line."""

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
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    from shrimpy_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(args.seed)
    N = 8
    big = np.ascontiguousarray(rng.random(4 * N * N * N * 4).astype(np.float32))      # generous: every shape <= N^3 fits
    out = np.empty_like(big)
    aux = np.ones_like(big)
    u16 = (big * 1000).astype(np.uint16)
    taps = np.ascontiguousarray(rng.random(64).astype(np.float32))
    table = np.ones(20 * 20 * 20, np.float64)
    d4 = np.empty(4, np.float64)
    cnt = np.empty(5000, np.uint32)
    idx = np.empty(1, np.int64)
    f = ctypes.c_float
    stats = {"calls": 0, "ok": 0, "refused": 0}

    def dim():
        return int(rng.choice([-3, 0, 1, 2, 3, 5, N, N, N - 1]))

    def ntap():
        return int(rng.choice([-1, 0, 1, 2, 3, 5, 7, 15, 17]))

    t_end = time.perf_counter() + args.seconds
    lib.lsr_set_host_threads(3)
    while time.perf_counter() < t_end:
        z, y, x, zo, yo, xo = dim(), dim(), dim(), dim(), dim(), dim()
        m = (ctypes.c_double * 12)(*[float(v) for v in rng.choice([0.0, 1.0, -1.0, 0.5, 2.0, float("nan"), 1e300], 12)])
        shear = (ctypes.c_double * 12)(-0.65, 0.0, 0.755, float(rng.integers(-3, 9)), -1.0, 0.0, 0.0, float(rng.integers(-2, 9)),
                                       0.0, -1.0, 0.0, float(rng.integers(-2, 9)))
        which = int(rng.integers(0, 12))
        if which == 0:
            rc = lib.lsr_deskew_f32_cpu(big.ctypes.data, z, y, x, out.ctypes.data, zo, yo, xo, int(rng.choice([xo, xo + 4, xo - 1, 0])),
                                        int(rng.choice([yo * xo if yo > 0 and xo > 0 else 0, 1, 4 * N * N])), dim(),
                                        shear if rng.random() < 0.7 else m, int(rng.integers(-1, 19)), None)
        elif which == 1:
            rc = lib.lsr_deskew_u16_cpu(u16.ctypes.data, z, y, x, out.ctypes.data, zo, yo, xo, max(xo, 0), max(yo, 0) * max(xo, 0), dim(),
                                        shear, int(rng.integers(-1, 19)), None)
        elif which == 2:
            rc = lib.lsr_affine_f32_cpu(big.ctypes.data, z, y, x, out.ctypes.data, zo, yo, xo, m, f(1.5), int(rng.integers(-1, 5)), None)
        elif which == 3:
            a = int(rng.integers(-1, 19))
            rc = lib.lsr_average_slices_f32_cpu(big.ctypes.data, z, y, x, out.ctypes.data, zo if rng.random() < 0.5 else -(-max(z, 1) // max(a, 1)), a, None)
        elif which == 4:
            rc = lib.lsr_correlate_sep_f32_cpu(big.ctypes.data, out.ctypes.data, aux.ctypes.data, z, y, x, taps.ctypes.data, ntap(),
                                               taps.ctypes.data, ntap(), taps.ctypes.data, ntap(), int(rng.integers(-1, 5)), f(1e-6),
                                               aux.ctypes.data, aux.ctypes.data, aux.ctypes.data, None)
        elif which == 5:
            rc = lib.lsr_correlate_dense_f32_cpu(big.ctypes.data, out.ctypes.data, aux.ctypes.data, z, y, x, big.ctypes.data, ntap(), ntap(),
                                                 ntap(), int(rng.integers(-1, 5)), f(1e-6), table.ctypes.data, None)
        elif which == 6:
            rc = lib.lsr_minmax_f32_cpu(big.ctypes.data, int(rng.choice([-1, 0, 1, 7, 1000])), d4.ctypes.data, None, None)
        elif which == 7:
            rc = lib.lsr_histogram_f32_cpu(big.ctypes.data, int(rng.choice([-1, 0, 1, 1000, 1 << 33])), f(float(rng.choice([0.0, 1.0]))),
                                           f(float(rng.choice([0.0, 1.0, 0.5]))), int(rng.choice([-1, 0, 1, 256, 4096, 5000])), cnt.ctypes.data, None)
        elif which == 8:
            fn = lib.lsr_weighted_centroid_f32_cpu if rng.random() < 0.5 else lib.lsr_mask_centroid_f32_cpu
            rc = fn(big.ctypes.data, z, y, x, f(0.3), d4.ctypes.data, None, None)
        elif which == 9:
            rc = lib.lsr_blur_reflect_f32_cpu(big.ctypes.data, out.ctypes.data, z, y, x, int(rng.integers(-1, 4)), taps.ctypes.data,
                                              int(rng.choice([-1, 0, 1, 2, 7, 8, 31, 64, 65])), f(0.1), f(float(rng.choice([0.0, 2.0]))), None)
        elif which == 10:
            rc = lib.lsr_match_shape_f32_cpu(big.ctypes.data, z, y, x, out.ctypes.data, zo, yo, xo, None)
        else:
            rc = lib.lsr_peak_abs_shifted_f32_cpu(big.ctypes.data, z, y, x, idx.ctypes.data, None, None)
        stats["calls"] += 1
        stats["ok" if rc == 0 else "refused"] += 1
    print(json.dumps(stats))


if __name__ == "__main__":
    main()
