"""Deskew and registration kernels away from the benchmark's one geometry: where the rate leaves the headline's.

    python tools/sweep_geometry.py [--reps 5]          # one JSON object per case

Deskew: light-sheet angle, pixel / scan ratio, slice averaging, overhang, uint16 / float32 input on a raw
(1024, 256, 2048) stack.  Registration: translations, the config-3 matrix, quarter turns and flips (axis swaps),
zooms, a large in-plane rotation and a 10 degree tilt on a (256, 1024, 2048) volume -- with the kernel each takes
(lsr_affine_path: 1 planar, 2 box, 0 gather).
"""

from __future__ import annotations

import argparse
import json
import sys

from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def timed(fn, reps):
    import torch

    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()

    import torch

    from shrimpy_amd import _lib
    from shrimpy_amd.deskew import fast_deskew_zyx, get_deskewed_data_shape
    from shrimpy_amd.geometry import as_matrix_3x4
    from shrimpy_amd.register import apply_affine_transform_zyx

    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(11)
    raw_shape = (1024, 256, 2048)
    raw = torch.randint(80, 4000, raw_shape, device=dev, generator=g, dtype=torch.int32)
    raw16, raw32 = raw.to(torch.uint16), raw.to(torch.float32)
    del raw
    base = dict(ls_angle_deg=30.0, px_to_scan_ratio=0.755, keep_overhang=False, average_n_slices=3)
    cases = [base, dict(base, average_n_slices=1), dict(base, average_n_slices=2), dict(base, average_n_slices=5),
             dict(base, keep_overhang=True), dict(base, ls_angle_deg=45.0), dict(base, ls_angle_deg=20.0),
             dict(base, px_to_scan_ratio=0.5), dict(base, px_to_scan_ratio=1.0), dict(base, px_to_scan_ratio=0.3),
             dict(base, px_to_scan_ratio=1.5)]
    for kw in cases:
        try:
            oshape = get_deskewed_data_shape(raw_shape, **kw)[0]
        except Exception as exc:  # noqa: BLE001 -- a geometry with no output is a finding, not a crash
            print(json.dumps({"kernel": "deskew", **kw, "error": str(exc)}), flush=True)
            continue
        n_in, n_o = raw16.numel(), int(np.prod(oshape))
        for name, src, bpe in (("uint16", raw16, 2), ("float32", raw32, 4)):
            ms = timed(lambda: fast_deskew_zyx(raw_data=src, **kw), args.reps)
            nbytes = bpe * n_in + 4.0 * n_o
            print(json.dumps({"kernel": "deskew", "input": name, **kw, "raw": raw_shape, "out": list(oshape), "ms": ms,
                              "algorithmic_GBps": nbytes / ms / 1e6, "frac_of_8TBps": nbytes / ms / 1e6 / 8000}), flush=True)
    del raw16, raw32
    torch.cuda.empty_cache()

    shape = (256, 1024, 2048)
    vol = torch.rand(shape, device=dev, generator=g) * 1000
    out = torch.empty_like(vol)

    def rot(axis, deg):
        c, s = np.cos(np.deg2rad(deg)), np.sin(np.deg2rad(deg))
        m = np.eye(4)
        a, b = [(1, 2), (0, 2), (0, 1)][axis]
        m[a, a], m[a, b], m[b, a], m[b, b] = c, -s, s, c
        return m

    def about_centre(m):
        centre = np.eye(4)
        centre[:3, 3] = [(n - 1) / 2 for n in shape]
        back = np.eye(4)
        back[:3, 3] = -centre[:3, 3]
        return centre @ m @ back

    def shift(*t):
        m = np.eye(4)
        m[:3, 3] = t
        return m

    th = np.deg2rad(2.0)
    config3 = np.eye(4)
    config3[:3, :3] = np.array([[1, 0, 0], [0, np.cos(th), -np.sin(th)], [0, np.sin(th), np.cos(th)]]) @ np.diag([1.0, 0.98, 1.02])
    config3[:3, 3] = [3.5, -12.25, 20.75]
    flip_x = np.diag([1.0, 1.0, -1.0, 1.0])
    flip_x[2, 3] = shape[2] - 1
    matrices = [("identity", np.eye(4)), ("integer shift", shift(2, -7, 13)), ("fractional shift", shift(0.5, -7.25, 13.125)),
                ("config 3", config3), ("flip x", flip_x), ("zoom 2x (samples every 0.5)", about_centre(np.diag([0.5, 0.5, 0.5, 1.0]))),
                ("decimate 2x in the plane", about_centre(np.diag([1.0, 2.0, 2.0, 1.0]))),
                ("decimate 2x in the plane, half-size target", np.diag([1.0, 2.0, 2.0, 1.0])),
                ("decimate 2x along z", about_centre(np.diag([2.0, 1.0, 1.0, 1.0]))),
                ("rotate 30 deg in the plane", about_centre(rot(0, 30.0))), ("rotate 10 deg in the plane", about_centre(rot(0, 10.0))),
                ("tilt 10 deg about y", about_centre(rot(1, 10.0))), ("tilt 5 deg about x", about_centre(rot(2, 5.0))),
                ("tilt 3 deg about y and x", about_centre(rot(1, 3.0) @ rot(2, 3.0)))]
    for name, m in matrices:
        for mode in ("constant", "grid-constant"):
            path = _lib.call_value("lsr_affine_path", *shape, _lib.matrix12(as_matrix_3x4(m)),
                                   _lib.MODE_CONSTANT if mode == "constant" else _lib.MODE_GRID_CONSTANT)
            half = name.endswith("half-size target")
            target = out[:, :shape[1] // 2, :shape[2] // 2].contiguous() if half else out
            for exact in (True, False):
                ms = timed(lambda: apply_affine_transform_zyx(vol, m, tuple(target.shape), mode=mode, out=target, exact=exact), args.reps)
                nbytes = 4.0 * vol.numel() + 4.0 * target.numel()
                print(json.dumps({"kernel": "affine apply", "map": name, "mode": mode, "arithmetic": "exact fp64" if exact else "f32",
                                  "path": {1: "planar", 2: "box", 0: "gather"}[path], "shape": shape, "ms": ms,
                                  "algorithmic_GBps": nbytes / ms / 1e6, "frac_of_8TBps": nbytes / ms / 1e6 / 8000}), flush=True)


if __name__ == "__main__":
    main()
