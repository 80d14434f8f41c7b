"""Read / write rate of one config-4 raw volume in the acquisition's layout (Zarr v3, one shard per
volume around blosc-zstd chunks, byte shuffle) into a pinned slot, per blosc backend (each in its own
process): "python" = the pure-Python frame walker, "auto" = the best one loadable (libblosc when
LSR_LIBBLOSC points at one, else this package's native walker).

    python tools/compressed_read_rate.py [--scratch /dev/shm] [--backends python,auto]
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def child(scratch, out_json):
    import numpy as np
    import torch

    import bench
    from shrimpy_amd.io import codecs
    from shrimpy_amd.io.omezarr import open_ome_zarr

    dev = torch.device("cuda:0")
    raw_shape = bench.WORKLOADS["config4"][1]
    vol = bench.synthetic_raw(raw_shape, 4000, dev).to(torch.uint16).cpu().numpy()
    root = Path(tempfile.mkdtemp(prefix="lsr_cz_", dir=scratch))
    res = {"backend": codecs.blosc_backend(), "zstd": None}
    try:
        with open_ome_zarr(root / "in.zarr", layout="hcs", mode="w", channel_names=["LS"], prefer_iohub=False,
                           version="0.5") as plate:
            arr = plate.create_position("A", "1", "0").create_zeros(
                "0", shape=(1, 1) + tuple(raw_shape), dtype="uint16", compress="blosc-zstd", shards="volume")
            t0 = time.perf_counter(); arr.write_volume(0, 0, vol); res["write_s"] = time.perf_counter() - t0
        nbytes = sum(f.stat().st_size for f in (root / "in.zarr").rglob("*") if f.is_file())
        res["stored_bytes"], res["raw_bytes"] = nbytes, vol.nbytes
        pinned = torch.empty(raw_shape, dtype=torch.uint16, pin_memory=True).numpy()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); arr.read_volume(0, 0, out=pinned); ts.append(time.perf_counter() - t0)
        res["read_s"] = ts
        assert (pinned == vol).all()
        res["zstd"] = codecs._zstd.name
        # one inner chunk, one thread: zstd alone against the whole frame
        one = np.ascontiguousarray(vol[:32])
        frame = codecs.blosc_encode(one, typesize=2, cname="zstd", clevel=1, shuffle=1)
        t0 = time.perf_counter(); codecs.blosc_decode(frame, out=np.empty_like(one)); res["one_chunk_decode_s"] = time.perf_counter() - t0
        res["one_chunk_bytes"] = one.nbytes
        print(json.dumps(res), file=open(out_json, "w"))
    finally:
        shutil.rmtree(root, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scratch", default=None)
    ap.add_argument("--backends", default="python,auto")
    ap.add_argument("--child", default=None)
    args = ap.parse_args()
    if args.child:
        return child(args.scratch, args.child)
    for b in args.backends.split(","):
        env = dict(os.environ)
        if b == "python":
            env["LSR_BLOSC"] = "python"
        out = tempfile.mktemp(suffix=".json")
        subprocess.run([sys.executable, __file__, "--child", out] + (["--scratch", args.scratch] if args.scratch else []),
                       env=env, check=True)
        d = json.load(open(out))
        d["requested"] = b
        d["read_GBps"] = d["raw_bytes"] / min(d["read_s"]) / 1e9
        d["write_GBps"] = d["raw_bytes"] / d["write_s"] / 1e9
        print(json.dumps(d))


if __name__ == "__main__":
    main()
