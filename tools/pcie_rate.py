"""PCIe-inclusive rate of the config-2 unit: host buffer in, host buffer out.

Three ways to hand a (2048, 512, 2048) stack over and get the (171, 2048, 2270) result back:
  reference   float32 from pageable memory, ``torch.as_tensor(..., device)`` inside the step,
              ``.cpu()`` after it (what shrimpy/preprocessing.py:316 does)
  u16         the camera's uint16 counts from pageable memory, same synchronous calls
  staged      uint16 through ``staging.VolumeStager`` (pinned slots, copy streams beside the kernels)
Prints one JSON line per mode: seconds per unit and voxels per second.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    import bench
    from shrimpy_amd.pipeline import VolumeReconstructor, run_sharded
    from shrimpy_amd.staging import VolumeStager

    ap = argparse.ArgumentParser()
    ap.add_argument("--units", type=int, default=6)
    ap.add_argument("--workload", default="config2")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    raw_shape = bench.WORKLOADS[args.workload][1]
    settings = bench.reconstruct_settings() if hasattr(bench, "reconstruct_settings") else None
    if settings is None:
        from shrimpy_amd.settings import DeconvolveSettings, DeskewSettings, ReconstructSettings

        settings = ReconstructSettings(
            deskew=DeskewSettings(pixel_size_um=0.1133, average_n_slices=bench.DESKEW["average_n_slices"],
                                  ls_angle_deg=bench.DESKEW["ls_angle_deg"],
                                  px_to_scan_ratio=bench.DESKEW["px_to_scan_ratio"],
                                  keep_overhang=bench.DESKEW["keep_overhang"]),
            deconvolution=DeconvolveSettings(iterations=bench.RL_ITERS, gaussian_shape_zyx=bench.PSF_SHAPE,
                                             gaussian_sigma_zyx=bench.PSF_SIGMA))
    rec = VolumeReconstructor(raw_shape, settings, dev)
    nvox, nraw = int(np.prod(rec.output_shape)), int(np.prod(raw_shape))
    rng = np.random.default_rng(0)
    raw16 = rng.integers(90, 900, raw_shape, dtype=np.uint16)
    rec(raw16)
    torch.cuda.synchronize()

    def line(mode, seconds, n, **kw):
        print(json.dumps({"mode": mode, "units": n, "s_per_unit": seconds / n,
                          "raw_voxels_per_s": nraw * n / seconds,       # bench.py's unit
                          "deskewed_voxels_per_s": nvox * n / seconds, **kw}), flush=True)

    # kernels only (input resident), for scale
    d16 = torch.as_tensor(raw16, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.units):
        rec(d16)
    torch.cuda.synchronize()
    line("resident (kernels only)", time.perf_counter() - t0, args.units)
    del d16

    for mode, host in (("u16 pageable, synchronous", raw16),
                       ("reference: float32 pageable, synchronous", raw16.astype(np.float32))):
        n = max(2, args.units // 2)
        t0 = time.perf_counter()
        for _ in range(n):
            res = rec(host).cpu()
        line(mode, time.perf_counter() - t0, n, host_GB=host.nbytes / 1e9)
        del res, host

    # the two halves of the synchronous hand-over, separately
    torch.as_tensor(raw16, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        d16 = torch.as_tensor(raw16, device=dev)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(json.dumps({"copy": "torch.as_tensor(pageable uint16 stack)", "GB": raw16.nbytes / 1e9, "s": dt,
                      "GBps": raw16.nbytes / dt / 1e9}), flush=True)
    res = rec(d16)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        host = res.cpu()
    dt = (time.perf_counter() - t0) / 3
    print(json.dumps({"copy": "result.cpu() (fresh pageable array each time)", "GB": host.numel() * 4 / 1e9, "s": dt,
                      "GBps": host.numel() * 4 / dt / 1e9}), flush=True)
    del d16, res, host

    stager = VolumeStager(raw_shape, np.uint16, rec.output_shape, dev)
    for k in range(stager.depth):
        stager.host_in(k)[...] = raw16          # camera frames land in the pinned slots
    sink = []
    t0 = time.perf_counter()
    run_sharded(list(range(args.units)), lambda u, out=None: out, rec, lambda u, v: sink.append(float(v[0, 0, 0])),
                synchronize=torch.cuda.synchronize, stager=stager)
    line("staged: u16 pinned slots, copy streams", time.perf_counter() - t0, args.units)


if __name__ == "__main__":
    main()
