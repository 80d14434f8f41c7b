"""Per-kernel timings that bench.py does not cover: affine apply (BASELINE config 3), the dense-PSF
RL launch, and the deskew kernel alone.  Prints one JSON object per kernel.

    python tools/bench_kernels.py [--reps 5]
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
    ap.add_argument("--skip-dense", action="store_true")
    ap.add_argument("--only-rl", action="store_true", help="skip the affine and deskew sections")
    ap.add_argument("--only-affine", action="store_true", help="only the affine apply section")
    ap.add_argument("--rl-grid", default="171,2048,2270", help="Z,Y,X of the RL launch section")
    ap.add_argument("--psf-sweep", action="store_true", help="only: ms per RL iteration over separable PSF sizes")
    ap.add_argument("--rl-fft", action="store_true", help="only: ms per RL iteration in the Fourier domain (dense PSFs "
                    "beyond the stencil kernels) next to the generic / dense stencils where they exist")
    ap.add_argument("--psf-sweep-wide", action="store_true", help="with --psf-sweep: every pz for in-plane extents 9-15")
    args = ap.parse_args()

    import torch

    import bench
    from shrimpy_amd.deconvolve import RichardsonLucyPlan
    from shrimpy_amd.deskew import deskew_with_matrix
    from shrimpy_amd.geometry import deskew_geometry
    from shrimpy_amd.register import apply_affine_transform_zyx

    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)

    if args.rl_fft:
        _rl_fft(args, torch, dev, g, tuple(int(v) for v in args.rl_grid.split(",")))
        return
    if args.psf_sweep:
        _psf_sweep(args, torch, dev, g, RichardsonLucyPlan, tuple(int(v) for v in args.rl_grid.split(",")))
        return
    if not args.only_rl:
        _affine_and_deskew(args, torch, dev, g, bench, deskew_with_matrix, deskew_geometry, apply_affine_transform_zyx)

    if args.only_affine:
        return
    # ---- RL launches on the config-2 grid: separable (tuned) and dense
    oshape = tuple(int(v) for v in args.rl_grid.split(","))
    if not args.only_rl:
        _estimators(args, torch, dev, g, oshape)
    _rl(args, torch, dev, g, bench, RichardsonLucyPlan, oshape)


def _affine_and_deskew(args, torch, dev, g, bench, deskew_with_matrix, deskew_geometry, apply_affine_transform_zyx):
    # ---- affine apply, config 3: 2048 x 2048 x 256 volume, rotation 2 deg o scale o translation
    shape = (256, 2048, 2048)
    vol = torch.rand(shape, device=dev, generator=g) * 1000
    out = torch.empty_like(vol)
    th = np.deg2rad(2.0)
    rot = np.array([[1, 0, 0], [0, np.cos(th), -np.sin(th)], [0, np.sin(th), np.cos(th)]])
    m = np.eye(4)
    m[:3, :3] = rot @ np.diag([1.0, 0.98, 1.02])
    m[:3, 3] = [3.5, -12.25, 20.75]
    for mode, exact in (("constant", True), ("grid-constant", True), ("constant", False)):
        ms = timed(lambda: apply_affine_transform_zyx(vol, m, mode=mode, out=out, exact=exact), args.reps)
        nbytes = 8.0 * vol.numel()
        print(json.dumps({"kernel": f"affine_kernel ({mode}, {'exact fp64' if exact else 'f32 interp'})", "shape": shape, "ms": ms,
                          "algorithmic_GBps": nbytes / ms / 1e6, "frac_of_8TBps": nbytes / ms / 1e6 / 8000,
                          "voxels_per_s": vol.numel() / (ms * 1e-3)}))
    # maps that couple z with the plane: (a) tilt 1.5 deg about y; (b) the config-3 registration with
    # a 3 deg tilt about y on top (the oblique light-sheet <-> label-free case) -- affine_box.hip;
    # the same maps in grid-constant mode take the general gather kernel (the old path)
    from shrimpy_amd import _lib
    from shrimpy_amd.geometry import as_matrix_3x4

    tilt = np.eye(4)
    c, sn = np.cos(np.deg2rad(1.5)), np.sin(np.deg2rad(1.5))
    tilt[0, 0], tilt[0, 2], tilt[2, 0], tilt[2, 2] = c, -sn, sn, c
    tilt[:3, 3] = [2.0, 0.5, -3.25]
    c3, s3 = np.cos(np.deg2rad(3.0)), np.sin(np.deg2rad(3.0))
    ry = np.array([[c3, 0, -s3], [0, 1, 0], [s3, 0, c3]])
    both = m.copy()
    both[:3, :3] = ry @ m[:3, :3]
    for name, mat in (("tilt 1.5deg about y", tilt), ("config3 o tilt 3deg about y", both)):
        for mode, exact in (("constant", True), ("constant", False), ("grid-constant", True)):
            path = _lib.call_value("lsr_affine_path", shape[0], shape[1], shape[2],
                                   _lib.matrix12(as_matrix_3x4(mat)), _lib.MODE_CONSTANT if mode == "constant" else _lib.MODE_GRID_CONSTANT)
            ms = timed(lambda: apply_affine_transform_zyx(vol, mat, mode=mode, out=out, exact=exact), args.reps)
            nbytes = 8.0 * vol.numel()
            print(json.dumps({"kernel": f"affine ({name}, {mode}, {'exact fp64' if exact else 'f32 interp'})",
                              "path": {0: "gather", 1: "planar", 2: "box"}[path],
                              "shape": shape, "ms": ms, "algorithmic_GBps": nbytes / ms / 1e6,
                              "frac_of_8TBps": nbytes / ms / 1e6 / 8000}))
    del vol, out
    if args.only_affine:
        return

    # ---- flat-field (median over Z + apply / fused deskew), config 2 raw stack of camera counts
    from shrimpy_amd.flatfield import flat_field_pattern

    raw_shape = bench.WORKLOADS["config2"][1]
    raw = torch.randint(80, 600, raw_shape, device=dev, generator=g).to(torch.float32)
    ms = timed(lambda: flat_field_pattern(raw), args.reps)             # camera counts: 2 passes
    print(json.dumps({"kernel": "flat_median_kernel (+ mean), integer counts", "raw": raw_shape, "ms": ms,
                      "passes_GBps": 2 * 4.0 * raw.numel() / ms / 1e6,
                      "frac_of_8TBps_at_2_passes": 2 * 4.0 * raw.numel() / ms / 1e6 / 8000}))
    raw += torch.rand(raw_shape, device=dev, generator=g)              # continuous values: 4 passes
    ms = timed(lambda: flat_field_pattern(raw), args.reps)
    print(json.dumps({"kernel": "flat_median_kernel (+ mean), float values", "raw": raw_shape, "ms": ms,
                      "passes_GBps": 4 * 4.0 * raw.numel() / ms / 1e6,
                      "frac_of_8TBps_at_4_passes": 4 * 4.0 * raw.numel() / ms / 1e6 / 8000}))
    ff = flat_field_pattern(raw)
    dst = torch.empty_like(raw)
    ms = timed(lambda: ff.apply(raw, out=dst), args.reps)
    print(json.dumps({"kernel": "flat_apply_kernel", "raw": raw_shape, "ms": ms,
                      "algorithmic_GBps": 8.0 * raw.numel() / ms / 1e6,
                      "frac_of_8TBps": 8.0 * raw.numel() / ms / 1e6 / 8000}))
    del dst
    geo = deskew_geometry(raw_shape, **bench.DESKEW)
    dsk = torch.empty(geo.output_shape, device=dev)
    for name, kw in (("deskew_kernel<false>, dense output rows", {}), ("deskew_kernel<true> (flat-field fused), dense output rows", {"flat_field": ff})):
        ms = timed(lambda: deskew_with_matrix(raw, geo.matrix_3x4, geo.pre_average_shape, 3, out=dsk, **kw), args.reps)
        print(json.dumps({"kernel": name, "raw": raw_shape, "ms": ms}))
    raw16 = raw.to(torch.uint16)
    ms = timed(lambda: deskew_with_matrix(raw16, geo.matrix_3x4, geo.pre_average_shape, 3, out=dsk), args.reps)
    print(json.dumps({"kernel": "deskew_kernel<false, U16> (uint16 camera counts in)", "raw": raw_shape, "ms": ms,
                      "algorithmic_GBps": (2.0 * raw16.numel() + 4.0 * dsk.numel()) / ms / 1e6}))
    ms = timed(lambda: flat_field_pattern(raw16), args.reps)
    print(json.dumps({"kernel": "flat_median_kernel<U16> (+ mean)", "raw": raw_shape, "ms": ms,
                      "passes_GBps": 2 * 2.0 * raw16.numel() / ms / 1e6}))
    ff16 = flat_field_pattern(raw16)
    ms = timed(lambda: deskew_with_matrix(raw16, geo.matrix_3x4, geo.pre_average_shape, 3, out=dsk, flat_field=ff16),
               args.reps)
    print(json.dumps({"kernel": "deskew_kernel<true, U16> (flat-field fused, uint16 in)", "raw": raw_shape, "ms": ms}))
    del raw, raw16, dsk, ff, ff16

    # ---- deskew alone, config 2 and config 4 mappings
    for name in ("config2", "config4"):
        raw_shape = bench.WORKLOADS[name][1]
        raw = torch.rand(raw_shape, device=dev, generator=g)
        geo = deskew_geometry(raw_shape, **bench.DESKEW)
        dst = torch.empty(geo.output_shape, device=dev)
        nbytes = 4.0 * raw.numel() + 4.0 * dst.numel()
        for border in ("constant", "grid-constant"):
            ms = timed(lambda: deskew_with_matrix(raw, geo.matrix_3x4, geo.pre_average_shape, 3, out=dst, border=border),
                       args.reps)
            # (dense output: rows of Xo floats, 2270 at config 2 -- not a multiple of four, so the stores of a tile row
            # straddle 16-byte boundaries; bench.py's step writes the RL plan's padded, line-aligned volume: 2.3 ms)
            print(json.dumps({"kernel": "deskew_kernel, dense output rows" + ("" if border == "constant" else " (border grid-constant)"),
                              "workload": name, "raw": raw_shape,
                              "out": geo.output_shape, "ms": ms, "algorithmic_GBps": nbytes / ms / 1e6,
                              "frac_of_8TBps": nbytes / ms / 1e6 / 8000}))
        del raw, dst


def _estimators(args, torch, dev, g, oshape):
    """DynaTrack estimators on a deskewed-size volume."""
    from shrimpy_amd import dynatrack as d

    vol = torch.rand(oshape, device=dev, generator=g) * 900 + 100
    n = vol.numel()
    for name, fn, passes in (
        ("_percentile (minmax + histogram)", lambda: d._percentile(vol, 90.0), 2),
        ("_intensity_center_of_mass", lambda: d._intensity_center_of_mass(vol, 150.0), 1),
        ("_gaussian_blur_3d sigma=2 (3 passes, 17 taps)", lambda: d._gaussian_blur_3d(vol, 2.0), 6),
        ("_gaussian_blur_3d sigma=5 (3 passes, 41 taps)", lambda: d._gaussian_blur_3d(vol, 5.0), 6),
        ("_multiotsu_center_of_mass sigma=2 (two volumes)", lambda: d._multiotsu_center_of_mass(vol, vol, 2.0), 20),
    ):
        ms = timed(fn, max(1, args.reps // 2))
        print(json.dumps({"kernel": name, "grid": oshape, "ms": ms, "volume_traversals": passes,
                          "traversal_GBps": passes * 4.0 * n / ms / 1e6}))
    # the reference's default tracker (dynatrack_demo.yaml:107): phase cross-correlation
    mov = torch.roll(vol, shifts=(2, -5, 7), dims=(0, 1, 2))
    shift = d._phase_cross_corr(vol, mov)
    assert shift in ((2, -5, 7), (-2, 5, -7)), shift
    from shrimpy_amd import fft3

    for route, ok in (("x and z legs in this package's kernels, rocFFT along y (fft3.correlation_peak)", True),
                      ("torch.fft.rfftn / irfftn", False)):
        if ok and not fft3.available():
            continue
        d._axis_fft_ok[0] = ok
        d.set_spectrum_cache_bytes(0)
        assert d._phase_cross_corr(vol, mov) == shift
        ms = timed(lambda: d._phase_cross_corr(vol, mov), max(1, args.reps // 2))
        print(json.dumps({"kernel": f"_phase_cross_corr (2 forward + 1 inverse 3-D FFT; {route})", "grid": oshape, "ms": ms,
                          "rolled_by": [2, -5, 7], "found": list(shift)}))
        d.set_spectrum_cache_bytes(8 << 30)
        ms = timed(lambda: d._phase_cross_corr(vol, mov), max(1, args.reps // 2))
        print(json.dumps({"kernel": f"_phase_cross_corr, reference spectrum cached (1 forward + 1 inverse; {route})",
                          "grid": oshape, "ms": ms}))
    d._axis_fft_ok[0] = True
    d.set_spectrum_cache_bytes(0)
    ms = timed(lambda: torch.fft.rfftn(vol), max(1, args.reps // 2))
    print(json.dumps({"kernel": "torch.fft.rfftn alone (rocFFT)", "grid": oshape, "ms": ms}))
    if fft3.available():
        ms = timed(lambda: fft3.rfft3(vol), max(1, args.reps // 2))
        print(json.dumps({"kernel": "fft3.rfft3 alone (at the volume's own shape)", "grid": oshape, "ms": ms}))


def _rl(args, torch, dev, g, bench, RichardsonLucyPlan, oshape):
    y = torch.poisson(torch.full(oshape, 100.0, device=dev), generator=g)
    plans = [("separable 9x7x7, one launch per iteration", RichardsonLucyPlan(oshape, None, dev, psf_factors=bench.gaussian_factors()), 4),
             ("separable 9x7x7, ratio / update launches",
              RichardsonLucyPlan(oshape, None, dev, psf_factors=bench.gaussian_factors(), fused="never"), 4),
             ("rotated 9x7x7 as ky (x) kzx, one launch per iteration (the default)", RichardsonLucyPlan(oshape, bench.rotated_psf(), dev), 4),
             ("rotated 9x7x7 as ky (x) kzx, ratio / update launches",
              RichardsonLucyPlan(oshape, bench.rotated_psf(), dev, fused="never"), 4)]
    if not args.skip_dense:
        plans.append(("dense 9x7x7 (rotated)", RichardsonLucyPlan(oshape, bench.rotated_psf(), dev, separable="never"), 1))
    for name, plan, iters in plans:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        plan(y, iterations=1)
        torch.cuda.synchronize()
        plan(y, iterations=iters, events=ev)
        torch.cuda.synchronize()
        launches = {"fused": 1, "y-separable (fused)": 1, "y-separable (4 launches)": 4}.get(plan.path, 2) * iters
        ms = ev[0].elapsed_time(ev[1]) / launches
        nbytes = 12.0 * y.numel()
        taps = {"fused": 46, "separable": 23, "y-separable": 70, "y-separable (fused)": 140}.get(plan.path, 441)
        plan.release()
        print(json.dumps({"kernel": f"RL launch, {name}", "path": plan.path, "grid": oshape, "ms_per_launch": ms,
                          "algorithmic_GBps": nbytes / ms / 1e6, "frac_of_8TBps": nbytes / ms / 1e6 / 8000,
                          "fma_TFLOPs": 2.0 * taps * y.numel() / ms / 1e9}))


def _psf_sweep(args, torch, dev, g, RichardsonLucyPlan, oshape):
    """ms per RL iteration over separable Gaussian PSFs from 3x3x3 to 15x15x15: which path each takes and where
    the rate leaves the 9x7x7 headline's (12 algorithmic bytes per voxel and iteration in every case)."""
    y = torch.poisson(torch.full(oshape, 100.0, device=dev), generator=g)
    sizes = [(3, 3, 3), (5, 5, 5), (7, 7, 7), (9, 7, 7), (9, 9, 9), (11, 9, 9), (13, 11, 11), (15, 9, 9), (15, 11, 11),
             (15, 15, 15), (5, 15, 15), (15, 3, 3)]
    if args.psf_sweep_wide:     # the in-plane extents where the fused tile shrinks: which form wins per (pz, pyx)
        sizes = [(pz, pyx, pyx) for pyx in (9, 11, 13, 15) for pz in (3, 5, 7, 9, 11, 13, 15)]
    for size in sizes:
        factors = []
        for n in size:
            x = np.arange(n) - n // 2
            k = np.exp(-0.5 * (x / (n / 4.5)) ** 2)
            factors.append((k / k.sum()).astype(np.float32))
        for fused in ("auto", "never"):
            plan = RichardsonLucyPlan(oshape, None, dev, psf_factors=tuple(factors), fused=fused)
            if fused == "never" and plan.path == "fused":
                plan.release()
                continue
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            plan(y, iterations=1)
            torch.cuda.synchronize()
            plan(y, iterations=4, events=ev)
            torch.cuda.synchronize()
            ms = ev[0].elapsed_time(ev[1]) / 4
            path = plan.path
            plan.release()
            print(json.dumps({"kernel": "RL iteration, separable PSF", "psf": list(size), "requested": fused, "path": path,
                              "grid": oshape, "ms_per_iteration": ms, "algorithmic_GBps": 12.0 * y.numel() / ms / 1e6,
                              "frac_of_8TBps": 12.0 * y.numel() / ms / 1e6 / 8000}), flush=True)
            if fused == "auto" and path != "fused":
                break      # (the two-launch form was what ran)


def _rl_fft(args, torch, dev, g, oshape):
    """ms per RL iteration with both convolutions in the Fourier domain (shrimpy_amd/deconvolve_fft.py), over dense PSF
    extents from the stencil kernels' range to a measured bead patch; the stencil route beside it where one exists."""
    from shrimpy_amd.deconvolve import make_plan

    y = torch.poisson(torch.full(oshape, 100.0, device=dev), generator=g)
    rng = np.random.default_rng(4)
    for size in [(9, 7, 7), (11, 9, 9), (13, 13, 13), (15, 15, 15), (15, 19, 19), (21, 15, 15), (31, 37, 19)]:
        zz, yy, xx = np.meshgrid(*[np.arange(n) - n // 2 for n in size], indexing="ij")
        zr, xr = 0.9 * zz + 0.43 * xx, -0.43 * zz + 0.9 * xx
        w = np.exp(-0.5 * ((zr / (size[0] / 5)) ** 2 + (yy / (size[1] / 5)) ** 2 + (xr / (size[2] / 5)) ** 2))
        w = (w * (1 + 0.02 * rng.standard_normal(size))).clip(0).astype(np.float32)
        w /= w.sum()
        for method in ("fft", "direct"):
            if method == "direct" and (max(size) > 15 or size[0] * size[1] * size[2] > 2200):
                continue      # (refused by the stencil plan / minutes per iteration through the generic stencil)
            plan = make_plan(oshape, w, dev, separable="never", method=method)
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            plan(y, iterations=1)
            torch.cuda.synchronize()
            iters = 3 if plan.path != "generic" else 1
            plan(y, iterations=iters, events=ev)
            torch.cuda.synchronize()
            ms = ev[0].elapsed_time(ev[1]) / iters
            row = {"kernel": "RL iteration, dense PSF", "psf": list(size), "method": method, "path": plan.path, "grid": oshape,
                   "ms_per_iteration": ms}
            if plan.path == "fft":
                row["fft_grid"] = list(plan.grid)
            plan.release()
            del plan
            torch.cuda.empty_cache()
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
