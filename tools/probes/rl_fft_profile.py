"""One dense PSF through the Fourier-domain RL iteration on the config-2 grid, for rocprofv3 --kernel-trace --stats:
   cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rl_fft_stats -- python3 $R/tools/probes/rl_fft_profile.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from shrimpy_amd.deconvolve import make_plan  # noqa: E402

dev = torch.device("cuda:0")
shape = (171, 2048, 2270)
size = (15, 19, 19)
zz, yy, xx = np.meshgrid(*[np.arange(n) - n // 2 for n in size], indexing="ij")
w = np.exp(-0.5 * (((0.9 * zz + 0.43 * xx) / 3) ** 2 + (yy / 3.5) ** 2 + ((-0.43 * zz + 0.9 * xx) / 3.5) ** 2)).astype(np.float32)
w /= w.sum()
y = torch.poisson(torch.full(shape, 100.0, device=dev))
plan = make_plan(shape, w, dev, method="fft")
plan(y, iterations=int(os.environ.get("ITERS", "5")))
torch.cuda.synchronize()
