"""End-to-end rate of the plate path (BASELINE configs 4-5 at reduced count):
OME-Zarr plate in -> deskew + 20-iteration RL -> OME-Zarr plate out, through ``cli.run_store``.

Writes a synthetic plate of ``--positions`` uint16 stacks of the config-4 raw shape to ``--dir``
(uncompressed chunks (1, 1, 32, ny, nx), the acquisition's layout), runs the store through the
pipeline twice -- synchronous hand-over, then pinned staging slots with copy streams -- and prints
one JSON line per mode.  The input is in the page cache after it has been written, so this is the
rate of everything but the disk.
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import bench
    from shrimpy_amd.cli import run_store
    from shrimpy_amd.io.omezarr import open_ome_zarr
    from shrimpy_amd.settings import DeconvolveSettings, DeskewSettings, ReconstructSettings

    ap = argparse.ArgumentParser()
    ap.add_argument("--positions", type=int, default=6)
    ap.add_argument("--workload", default="config4")
    ap.add_argument("--dir", default=None)
    args = ap.parse_args()
    raw_shape = bench.WORKLOADS[args.workload][1]
    root = tempfile.mkdtemp(prefix="lsr_plate_", dir=args.dir)
    try:
        rng = np.random.default_rng(4000)
        base = rng.integers(90, 900, raw_shape, dtype=np.uint16)
        t0 = time.perf_counter()
        with open_ome_zarr(os.path.join(root, "raw.zarr"), layout="hcs", mode="w", channel_names=["LS"],
                           prefer_iohub=False) as plate:
            for i in range(args.positions):
                pos = plate.create_position("A", str(i + 1), "0")
                arr = pos.create_zeros("0", shape=(1, 1) + tuple(raw_shape), dtype="uint16",
                                       scale=(1, 1, 0.15, 0.1133, 0.1133))
                arr.write_volume(0, 0, base + np.uint16(i))
        print(json.dumps({"wrote": args.positions, "raw_shape": list(raw_shape),
                          "GB_per_position": base.nbytes / 1e9, "write_s": time.perf_counter() - t0}), flush=True)
        settings = ReconstructSettings(
            deskew=DeskewSettings(pixel_size_um=0.1133, ls_angle_deg=bench.DESKEW["ls_angle_deg"],
                                  px_to_scan_ratio=bench.DESKEW["px_to_scan_ratio"],
                                  keep_overhang=bench.DESKEW["keep_overhang"],
                                  average_n_slices=bench.DESKEW["average_n_slices"]),
            deconvolution=DeconvolveSettings(iterations=bench.RL_ITERS, gaussian_shape_zyx=bench.PSF_SHAPE,
                                             gaussian_sigma_zyx=bench.PSF_SIGMA))
        for mode, staged in (("synchronous hand-over", False), ("pinned slots + copy streams", True)):
            out = os.path.join(root, f"out_{int(staged)}.zarr")
            t0 = time.perf_counter()
            rep = run_store(os.path.join(root, "raw.zarr"), out, settings, stage_through_pinned=staged)
            dt = time.perf_counter() - t0
            nvox = args.positions * int(np.prod(raw_shape))
            print(json.dumps({"mode": mode, "units": rep["units"], "wall_s": dt, "job_s": rep["job_seconds"],
                              "s_per_unit": rep["job_seconds"] / rep["units"],
                              "raw_voxels_per_s": nvox / rep["job_seconds"],
                              "output_shape": list(rep["output_shape"])}), flush=True)
            shutil.rmtree(out, ignore_errors=True)
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
