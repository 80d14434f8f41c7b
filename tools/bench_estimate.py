"""Time the registration estimate on a config-3-like pair: python tools/bench_estimate.py [Z Y X]"""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

import bench
from shrimpy_amd.estimate import estimate_affine_zyx, normal_equations
from shrimpy_amd.register import apply_affine_transform_zyx

shape = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (128, 1024, 1024)
dev = torch.device("cuda:0")
mov = bench.synthetic_raw(shape, 3000, dev)
true = bench.registration_matrix()
c3, s3 = np.cos(np.deg2rad(1.0)), np.sin(np.deg2rad(1.0))
true[:3, :3] = np.array([[c3, 0, -s3], [0, 1, 0], [s3, 0, c3]]) @ true[:3, :3]
c = np.array([(n - 1) / 2 for n in shape])
true[:3, 3] = c - true[:3, :3] @ c + np.array([1.5, -4.25, 6.75])
tgt = 1.3 * apply_affine_transform_zyx(mov, true, shape) + 20.0
torch.cuda.synchronize()
for stride in (4, 2, 1):
    normal_equations(mov, tgt, np.eye(4), stride=stride)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        normal_equations(mov, tgt, np.eye(4), stride=stride)
    torch.cuda.synchronize()
    print(json.dumps({"kernel": "affine_normal_kernel", "shape": shape, "stride": stride,
                      "ms_per_launch_incl_host_sum": (time.perf_counter() - t0) / 3 * 1e3}))
t0 = time.perf_counter()
est = estimate_affine_zyx(mov, tgt)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
corners = np.array([[z, y, x, 1.0] for z in (0, shape[0] - 1) for y in (0, shape[1] - 1) for x in (0, shape[2] - 1)])
print(json.dumps({"estimate": "affine + intensity, default_levels(shape)", "shape": shape, "seconds": dt,
                  "iterations": est.iterations, "converged": est.converged, "rms": est.rms,
                  "max_corner_error_voxels": float(np.abs(corners @ (est.affine_transform_zyx - true).T).max()),
                  "gain": est.gain, "offset": est.offset}))
