"""Serial time of every stage of one unit on its way store -> GPU -> store (config-4 position by default):
chunk files -> pinned slot, upload, kernels, download, pinned slot -> chunk files.  The streamed run
(cli.run_store) overlaps them; its rate is set by the slowest line printed here.

    python tools/stage_times.py [--workload config4] [--scratch /dev/shm] [--reps 3]
"""
import argparse
import json
import shutil
import sys
import tempfile
import time

from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="config4")
    ap.add_argument("--scratch", default=None)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import numpy as np
    import torch

    import bench
    from shrimpy_amd.io.omezarr import open_ome_zarr
    from shrimpy_amd.pipeline import VolumeReconstructor
    from shrimpy_amd.staging import VolumeStager

    dev = torch.device("cuda:0")
    cid, raw_shape = bench.WORKLOADS[args.workload]
    settings = bench.plate_settings(args.workload)
    rec = VolumeReconstructor(raw_shape, settings, dev)
    oshape = rec.output_shape
    root = Path(tempfile.mkdtemp(prefix="lsr_stage_", dir=args.scratch))
    try:
        with open_ome_zarr(root / "in.zarr", layout="hcs", mode="w", channel_names=["LS"], prefer_iohub=False) as plate:
            src = plate.create_position("A", "1", "0").create_zeros("0", shape=(1, 1) + tuple(raw_shape), dtype="uint16")
            raw = bench.synthetic_raw(raw_shape, 1000 * cid, dev).to(torch.uint16)
            src.write_volume(0, 0, raw.cpu().numpy())
            from shrimpy_amd.io.omezarr import create_level
            dst = create_level(plate.create_position("A", "2", "0"), (1, 1) + tuple(oshape), "float32", (1, 1, 1, 1, 1))
        st = VolumeStager(raw_shape, np.uint16, oshape, dev)
        t = {k: [] for k in ("read_chunks_to_pinned", "upload", "kernels", "download", "write_chunks_from_pinned")}
        for _ in range(args.reps):
            t0 = time.perf_counter(); src.read_volume(0, 0, out=st.host_in(0)); t["read_chunks_to_pinned"].append(time.perf_counter() - t0)
            torch.cuda.synchronize(); t0 = time.perf_counter(); st.stage_in(0); d = st.acquire(0); torch.cuda.synchronize()
            t["upload"].append(time.perf_counter() - t0)
            t0 = time.perf_counter(); out = rec(d); torch.cuda.synchronize(); t["kernels"].append(time.perf_counter() - t0)
            st.release(0)
            t0 = time.perf_counter(); st.stage_out(0, out); host = st.collect(0); t["download"].append(time.perf_counter() - t0)
            t0 = time.perf_counter(); dst.write_volume(0, 0, host); t["write_chunks_from_pinned"].append(time.perf_counter() - t0)
        nb_in, nb_out = int(np.prod(raw_shape)) * 2, int(np.prod(oshape)) * 4
        print(json.dumps({"workload": args.workload, "scratch": str(root.parent), "bytes_in": nb_in, "bytes_out": nb_out,
                          "seconds_min": {k: min(v) for k, v in t.items()},
                          "GBps": {"read": nb_in / min(t["read_chunks_to_pinned"]) / 1e9, "upload": nb_in / min(t["upload"]) / 1e9,
                                   "download": nb_out / min(t["download"]) / 1e9,
                                   "write": nb_out / min(t["write_chunks_from_pinned"]) / 1e9}}))
        st.close()
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
