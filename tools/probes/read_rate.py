"""Read rate of config-4 raw volumes from chunk files into a pinned slot: first touch of each
position against repeats, alone and with a concurrent writer / concurrent PCIe copies."""
import json
import shutil
import sys
import tempfile
import threading
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
import torch

from shrimpy_amd.io.omezarr import create_level, open_ome_zarr
from shrimpy_amd.staging import VolumeStager

raw_shape, oshape = (2048, 256, 2048), (86, 2048, 2491)
dev = torch.device("cuda:0")
root = Path(tempfile.mkdtemp(prefix="lsr_rr_", dir=sys.argv[1] if len(sys.argv) > 1 else "/dev/shm"))
try:
    n = 6
    src = []
    with open_ome_zarr(root / "in.zarr", layout="hcs", mode="w", channel_names=["LS"], prefer_iohub=False) as plate:
        v = (np.random.default_rng(0).integers(0, 4000, raw_shape, dtype=np.uint16))
        for p in range(n):
            a = plate.create_position("A", str(p + 1), "0").create_zeros("0", shape=(1, 1) + raw_shape, dtype="uint16")
            a.write_volume(0, 0, v)
            src.append(a)
        dst = [create_level(plate.create_position("B", str(p + 1), "0"), (1, 1) + oshape, "float32", (1,) * 5) for p in range(n)]
    st = VolumeStager(raw_shape, np.uint16, oshape, dev)
    res = {}

    def reads(tag):
        ts = []
        for p in range(n):
            t0 = time.perf_counter(); src[p].read_volume(0, 0, out=st.host_in(p % 2)); ts.append(round(time.perf_counter() - t0, 4))
        res[tag] = ts

    reads("alone_first")
    reads("alone_again")
    out_host = st._host_out[0].numpy()
    stop = threading.Event()

    def writer():
        k = 0
        while not stop.is_set():
            dst[k % n].write_volume(0, 0, out_host); k += 1
        res.setdefault("writes_done", []).append(k)

    th = threading.Thread(target=writer); th.start(); time.sleep(0.05)
    reads("with_writer")
    stop.set(); th.join(); stop.clear()

    d_in = st._dev_in[0]; d_out = torch.empty(oshape, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def dma():
        while not stop.is_set():
            with torch.cuda.stream(s1):
                d_in.copy_(st._host_in[1], non_blocking=True)
            with torch.cuda.stream(s2):
                st._host_out[1].copy_(d_out, non_blocking=True)
            s1.synchronize(); s2.synchronize()

    th = threading.Thread(target=dma); th.start(); time.sleep(0.05)
    ts = []
    for p in range(n):
        t0 = time.perf_counter(); src[p].read_volume(0, 0, out=st.host_in(0)); ts.append(round(time.perf_counter() - t0, 4))
    res["with_dma"] = ts
    th2 = threading.Thread(target=writer); th2.start(); time.sleep(0.05)
    ts = []
    for p in range(n):
        t0 = time.perf_counter(); src[p].read_volume(0, 0, out=st.host_in(0)); ts.append(round(time.perf_counter() - t0, 4))
    res["with_dma_and_writer"] = ts
    stop.set(); th.join(); th2.join()
    # writes alone, fresh files vs overwrite
    ts = []
    for p in range(n):
        t0 = time.perf_counter(); dst[p].write_volume(0, 0, out_host); ts.append(round(time.perf_counter() - t0, 4))
    res["write_overwrite"] = ts
    print(json.dumps(res))
    st.close()
finally:
    shutil.rmtree(root, ignore_errors=True)
