"""The Fourier-domain RL iteration at the deskewed sizes of BASELINE configs 4 and 5 (other radix mixes along x than
config 2's 1152 = 2^7 3^2: 1250 = 2 5^4 and 1280 = 2^8 5), two iterations with a bead-patch PSF, checked against the
oracle on crops with full margin.  python tools/probes/rl_fft_sizes.py"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from oracle import cpu_ref as o  # noqa: E402
from shrimpy_amd.deconvolve import make_plan  # noqa: E402

dev = torch.device("cuda:0")
psf = bench.measured_psf((15, 19, 19))
for name, shape in (("config4", (86, 2048, 2491)), ("config5", (67, 2048, 2540)), ("odd", (33, 1001, 1777))):
    g = torch.Generator(device=dev).manual_seed(5)
    y = torch.poisson(torch.rand(shape, device=dev, generator=g) * 300 + 50, generator=g)
    plan = make_plan(shape, psf, dev)
    t0 = time.perf_counter()
    x = plan(y, iterations=2)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n, core = 2, 24
    my, mx = n * 2 * 9, n * 2 * 9
    worst = 0.0
    for (y0, x0) in ((my, mx), (shape[1] // 2, shape[2] // 2), (shape[1] - core - my, shape[2] - core - mx)):
        crop = y[:, y0 - my:y0 + core + my, x0 - mx:x0 + core + mx].contiguous().cpu().numpy()
        want = o.richardson_lucy(crop, psf, n, use_fft=True)[:, my:my + core, mx:mx + core].astype(np.float64)
        got = x[:, y0:y0 + core, x0:x0 + core].cpu().numpy().astype(np.float64)
        excess = np.abs(got - want) - (2e-4 * np.abs(want) + 1e-4 * np.abs(want).max())
        worst = max(worst, float(excess.max()))
    print(json.dumps({"case": name, "shape": shape, "fft_grid": list(plan.grid), "path": plan.path, "seconds_2_iterations": round(dt, 3),
                      "max_excess_over_rl_bar": worst, "ok": worst <= 0}), flush=True)
    plan.release()
    del plan, x, y
    torch.cuda.empty_cache()
