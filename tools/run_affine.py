"""Run the config-3 affine apply (planar kernel, exact and f32 modes) a few times (profiling target)."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from shrimpy_amd.register import apply_affine_transform_zyx

th = np.deg2rad(2.0)
m = np.eye(4)
m[:3, :3] = np.array([[1, 0, 0], [0, np.cos(th), -np.sin(th)], [0, np.sin(th), np.cos(th)]]) @ np.diag([1.0, 0.98, 1.02])
m[:3, 3] = [3.5, -12.25, 20.75]
g = torch.Generator(device="cuda").manual_seed(1)
vol = torch.rand((256, 2048, 2048), device="cuda", generator=g)
out = torch.empty_like(vol)
for exact in (True, False) if len(sys.argv) < 2 else (sys.argv[1] == "exact",):
    for _ in range(3):
        apply_affine_transform_zyx(vol, m, out=out, exact=exact)
torch.cuda.synchronize()
