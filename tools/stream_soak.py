"""Many small units through the streamed store-to-store path, every output checked.

    python tools/stream_soak.py [--seconds 60] [--dir /dev/shm]

The staging slots (pinned host buffers, copy streams, events: shrimpy_amd/staging.py) are reused every `depth`
units; with full-size volumes a run has a dozen units and every stage takes tens of milliseconds.  Here the volumes
are small and many (hundreds of units per plate, a few milliseconds each, sizes and iteration counts varied per
round), so the loader, the kernels, the downloads and the writer chase each other closely -- and every volume of
the output store is compared bit for bit with the same unit reconstructed on its own.  One JSON line per round.
"""

from __future__ import annotations

import argparse
import json
import shutil
import sys
import tempfile
import time

from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--dir", default=None)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--engine-format", action="store_true",
                    help="input stores in the acquisition's format (Zarr v3, one blosc-zstd shard per volume, 32 KB blocks): "
                         "with the CLI's default blosc-zstd output both device codecs are in the loop")
    args = ap.parse_args()

    import torch

    from shrimpy_amd import cli
    from shrimpy_amd.io.omezarr import as_volume_array, open_ome_zarr
    from shrimpy_amd.pipeline import VolumeReconstructor
    from shrimpy_amd.settings import DeconvolveSettings, DeskewSettings, ReconstructSettings

    dev = torch.device("cuda:0")
    rng = np.random.default_rng(args.seed)
    t_end = time.perf_counter() + args.seconds
    rnd = 0
    while time.perf_counter() < t_end:
        rnd += 1
        raw_shape = (int(rng.integers(40, 200)), int(rng.integers(6, 40)), int(rng.choice([64, 100, 128, 257, 512])))
        n_p, n_t = int(rng.integers(3, 9)), int(rng.integers(4, 40))
        iters = int(rng.integers(0, 6))
        dtype = "uint16" if rng.random() < 0.7 else "float32"
        settings = ReconstructSettings(
            deskew=DeskewSettings(pixel_size_um=0.1133, scan_step_um=0.15, ls_angle_deg=30.0,
                                  average_n_slices=int(rng.integers(1, 4))),
            deconvolution=None if iters == 0 else DeconvolveSettings(iterations=iters, gaussian_shape_zyx=(5, 5, 5),
                                                                     gaussian_sigma_zyx=(1.2, 1.0, 1.0)))
        root = Path(tempfile.mkdtemp(prefix="lsr_soak_", dir=args.dir))
        try:
            keys = [f"A/{p + 1}/0" for p in range(n_p)]
            fmt = dict(compress="blosc-zstd", shards="volume", blocksize=32768) if args.engine_format else {}
            with open_ome_zarr(root / "in.zarr", layout="hcs", mode="w", channel_names=["LS"], prefer_iohub=False,
                               version="0.5" if args.engine_format else "0.4") as plate:
                for p, key in enumerate(keys):
                    arr = plate.create_position(*key.split("/")).create_zeros(
                        "0", shape=(n_t, 1) + raw_shape, dtype=dtype, scale=(1, 1, 0.15, 0.1133, 0.1133), **fmt)
                    for t in range(n_t):
                        vol = np.random.default_rng(1000 * rnd + 50 * p + t).integers(90, 900, raw_shape)
                        arr.write_volume(t, 0, vol.astype(dtype))
            t0 = time.perf_counter()
            # (engine format: both device codecs forced -- left to itself the run would decode volumes this small on the host)
            res = cli.run_store(root / "in.zarr", root / "out.zarr", settings,
                                **(dict(compression="blosc-zstd", zarr_version="0.5", device_codec=True) if args.engine_format else {}))
            dt = time.perf_counter() - t0
            rec = VolumeReconstructor(raw_shape, settings, dev)
            bad = []
            with open_ome_zarr(root / "out.zarr", layout="hcs", mode="r", prefer_iohub=False) as out:
                positions = dict(out.positions())
                for p, key in enumerate(keys):
                    arr = as_volume_array(positions[key]["0"])
                    for t in range(n_t):
                        vol = np.random.default_rng(1000 * rnd + 50 * p + t).integers(90, 900, raw_shape).astype(dtype)
                        want = rec(torch.as_tensor(vol, device=dev)).cpu().numpy()
                        if not np.array_equal(arr.read_volume(t, 0), want):
                            bad.append((key, t))
            print(json.dumps({"round": rnd, "raw_shape": raw_shape, "dtype": dtype, "units": n_p * n_t, "rl_iterations": iters,
                              "seconds": round(dt, 3), "units_per_s": round(n_p * n_t / dt, 1), "mismatches": bad[:5],
                              "n_mismatches": len(bad), "device_codec": res.get("device_codec"),
                              "stage_seconds": res.get("stage_seconds")}), flush=True)
            if bad:
                raise SystemExit(1)
        finally:
            shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
