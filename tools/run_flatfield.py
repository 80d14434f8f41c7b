"""Run the flat-field median a few times on the config-2 raw shape (profiling target)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from shrimpy_amd.flatfield import flat_field_pattern

g = torch.Generator(device="cuda").manual_seed(1)
raw = torch.randint(80, 600, (2048, 512, 2048), device="cuda", generator=g).to(torch.float32)
raw += torch.rand(raw.shape, device="cuda", generator=g)
for _ in range(3):
    flat_field_pattern(raw)
torch.cuda.synchronize()
