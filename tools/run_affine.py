"""Run one config-3-size affine apply a few times (profiling target).

    python tools/run_affine.py [planar|tilt|both] [exact|f32|all] [constant|grid-constant]

planar = the config-3 registration (affine_planar.hip); tilt = 1.5 deg about y; both = config 3 with a
3 deg tilt about y on top (affine_box.hip)."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from shrimpy_amd.register import apply_affine_transform_zyx

which = sys.argv[1] if len(sys.argv) > 1 else "planar"
modes = (sys.argv[2] == "exact",) if len(sys.argv) > 2 and sys.argv[2] != "all" else (True, False)
border = sys.argv[3] if len(sys.argv) > 3 else "constant"
th = np.deg2rad(2.0)
m = np.eye(4)
m[:3, :3] = np.array([[1, 0, 0], [0, np.cos(th), -np.sin(th)], [0, np.sin(th), np.cos(th)]]) @ np.diag([1.0, 0.98, 1.02])
m[:3, 3] = [3.5, -12.25, 20.75]
if which == "tilt":
    m = np.eye(4)
    c, s = np.cos(np.deg2rad(1.5)), np.sin(np.deg2rad(1.5))
    m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c, -s, s, c
    m[:3, 3] = [2.0, 0.5, -3.25]
elif which == "both":
    c, s = np.cos(np.deg2rad(3.0)), np.sin(np.deg2rad(3.0))
    m[:3, :3] = np.array([[c, 0, -s], [0, 1, 0], [s, 0, c]]) @ m[:3, :3]
g = torch.Generator(device="cuda").manual_seed(1)
vol = torch.rand((256, 2048, 2048), device="cuda", generator=g)
out = torch.empty_like(vol)
for exact in modes:
    for _ in range(3):
        apply_affine_transform_zyx(vol, m, out=out, exact=exact, mode=border)
torch.cuda.synchronize()
