"""Where does a streamed store-to-store run spend its time?  Runs ``cli.run_store`` on a small plate
of config-4 (or config-5) units with the reader, the writer and the stager's wait points timed, and
prints one JSON line: per-stage busy seconds, the waits of each thread, and the run's s/unit.

    python tools/stream_timeline.py [--workload config4] [--units 12] [--scratch /dev/shm]
"""
import argparse
import json
import shutil
import sys
import tempfile
import threading
import time

from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="config4")
    ap.add_argument("--units", type=int, default=12)
    ap.add_argument("--scratch", default=None)
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--engine-format", action="store_true",
                    help="input as the acquisition writes it: Zarr v3, one shard per volume around blosc-zstd chunks "
                         "(frames made by c-blosc when LSR_LIBBLOSC points at one)")
    ap.add_argument("--read-backend", default=None, choices=["libblosc", "lsrecon", "python"],
                    help="who decodes the frames in the timed run (default: the best one loadable)")
    args = ap.parse_args()
    import torch

    import bench
    from shrimpy_amd import cli, pipeline, staging
    from shrimpy_amd.io import omezarr
    from shrimpy_amd.io.omezarr import open_ome_zarr

    dev = torch.device("cuda:0")
    cid, raw_shape = bench.WORKLOADS[args.workload]
    settings = bench.plate_settings(args.workload)
    root = Path(tempfile.mkdtemp(prefix="lsr_tl_", dir=args.scratch))
    spans = []            # (name, thread, t0, t1)
    lock = threading.Lock()
    t_origin = [0.0]

    def timed(name, fn):
        def wrapper(*a, **k):
            t0 = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                t1 = time.perf_counter()
                with lock:
                    spans.append((name, threading.current_thread().name, t0 - t_origin[0], t1 - t_origin[0]))
        return wrapper

    try:
        fmt = dict(compress="blosc-zstd", shards="volume") if args.engine_format else {}
        with open_ome_zarr(root / "in.zarr", layout="hcs", mode="w", channel_names=["LS"], prefer_iohub=False,
                           version="0.5" if args.engine_format else "0.4") as plate:
            for p in range(args.units):
                arr = plate.create_position("A", str(p + 1), "0").create_zeros(
                    "0", shape=(1, 1) + tuple(raw_shape), dtype="uint16", scale=(1, 1, 0.15, 0.1133, 0.1133), **fmt)
                v = bench.synthetic_raw(raw_shape, seed=1000 * cid + 7 * p, device=dev)
                arr.write_volume(0, 0, v.to(torch.uint16).cpu().numpy())
                del v
        torch.cuda.empty_cache()
        from shrimpy_amd.io import codecs
        writer = codecs.blosc_backend()
        if args.read_backend in ("lsrecon", "python"):
            codecs._libblosc, codecs._libblosc_tried = None, True
            codecs._numcodecs_blosc, codecs._numcodecs_tried = None, True
        if args.read_backend == "python":
            codecs._native, codecs._native_tried = None, True
        omezarr.ZarrArray.read_volume = timed("read", omezarr.ZarrArray.read_volume)
        omezarr.ZarrArray.write_volume = timed("write", omezarr.ZarrArray.write_volume)
        staging.VolumeStager.host_in = timed("wait_host_in", staging.VolumeStager.host_in)
        staging.VolumeStager.collect = timed("wait_download", staging.VolumeStager.collect)
        staging.VolumeStager.stage_out = timed("stage_out", staging.VolumeStager.stage_out)
        pipeline.VolumeReconstructor.__call__ = timed("launch", pipeline.VolumeReconstructor.__call__)
        t_origin[0] = time.perf_counter()
        res = cli.run_store(root / "in.zarr", root / "out.zarr", settings)
        total = time.perf_counter() - t_origin[0]
        busy = {}
        for name, _, a, b in spans:
            busy.setdefault(name, []).append(b - a)
        summary = {k: {"n": len(v), "sum": round(sum(v), 4), "median": round(sorted(v)[len(v) // 2], 4),
                       "max": round(max(v), 4)} for k, v in busy.items()}
        print(json.dumps({"workload": args.workload, "units": res["units_total"], "job_seconds": res["job_seconds"],
                          "wall_seconds": total, "s_per_unit": res["job_seconds"] / res["units_total"],
                          "scratch": str(root.parent), "engine_format": args.engine_format, "frames_written_by": writer if args.engine_format else None,
                          "blosc_backend": __import__("shrimpy_amd.io.codecs", fromlist=["x"]).blosc_backend(),
                          "stages": summary}))
        if args.verbose:
            for name, th, a, b in sorted(spans, key=lambda s: s[2]):
                print(f"{a:8.3f} {b:8.3f} {b - a:7.3f}  {name:14s} {th}")
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
