"""Run the config-2 deskew a few times (profiling target): python tools/run_deskew.py [f32|u16]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

import bench
from shrimpy_amd.deskew import deskew_with_matrix
from shrimpy_amd.geometry import deskew_geometry

kind = sys.argv[1] if len(sys.argv) > 1 else "f32"
shape = bench.WORKLOADS["config2"][1]
g = torch.Generator(device="cuda").manual_seed(1)
raw = torch.randint(80, 600, shape, device="cuda", generator=g, dtype=torch.int32)
raw = raw.to(torch.uint16) if kind == "u16" else raw.to(torch.float32)
geo = deskew_geometry(shape, **bench.DESKEW)
out = torch.empty(geo.output_shape, dtype=torch.float32, device="cuda")
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for i in range(6):
    if i == 1:
        ev[0].record()
    deskew_with_matrix(raw, geo.matrix_3x4, geo.pre_average_shape, 3, out=out)
ev[1].record()
torch.cuda.synchronize()
print(f"deskew {kind}: {ev[0].elapsed_time(ev[1]) / 5:.3f} ms per launch (dense output)")
