"""One cached-reference _phase_cross_corr on the config-2 deskewed grid, for a kernel trace."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from shrimpy_amd import dynatrack as d
dev = torch.device("cuda:0")
vol = torch.rand((171, 2048, 2270), device=dev)
mov = torch.roll(vol, shifts=(2, -5, 7), dims=(0, 1, 2))
d.set_spectrum_cache_bytes(8 << 30)
d._phase_cross_corr(vol, mov)
torch.cuda.synchronize()
torch.zeros(1, device=dev).add_(1); torch.cuda.synchronize()
print(d._phase_cross_corr(vol, mov))
torch.cuda.synchronize()
torch.zeros(1, device=dev).add_(2); torch.cuda.synchronize()
