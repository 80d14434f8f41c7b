"""Affine apply on a moving volume whose width is not a multiple of 4 (a deskewed config-2 volume is 2270 wide):
dense input (gather kernel), dense input with the automatic padded copy, and a PitchedVolume the deskew wrote."""
import json
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
import torch
import bench
from shrimpy_amd.register import PitchedVolume, apply_affine_transform_zyx

dev = torch.device("cuda:0")
shape = (171, 2048, 2270)
vol = torch.rand(shape, device=dev)
out = torch.empty(shape, device=dev)
m = bench.registration_matrix()


def timed(fn, reps=6):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


nbytes = 8.0 * vol.numel()
for exact in (True, False):
    t_gather = timed(lambda: apply_affine_transform_zyx(vol, m, shape, out=out, exact=exact))
    t_auto = timed(lambda: apply_affine_transform_zyx(PitchedVolume.copy_of(vol), m, shape, out=out, exact=exact))
    pv = PitchedVolume.copy_of(vol)
    t_pitched = timed(lambda: apply_affine_transform_zyx(pv, m, shape, out=out, exact=exact))
    print(json.dumps({"shape": shape, "exact": exact, "dense_gather_ms": t_gather, "padded_copy_then_pitched_ms": t_auto,
                      "pitched_ms": t_pitched, "pitched_frac_of_8TBps": nbytes / t_pitched / 1e6 / 8000}))
