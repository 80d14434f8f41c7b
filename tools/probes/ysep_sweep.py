"""ms per RL iteration of the one-launch ky (x) kzx kernel against the two-launch form over PSF extents (config-2 grid):
python tools/probes/ysep_sweep.py  ->  one JSON line per extent (profiles/r04_ysep_sweep.jsonl)."""
import json, sys
sys.path.insert(0, '.')
import numpy as np, torch
from shrimpy_amd.deconvolve import RichardsonLucyPlan
dev = torch.device("cuda:0")
shape = (171, 2048, 2270)
g = torch.Generator(device=dev).manual_seed(5)
y = torch.poisson(torch.full(shape, 100.0, device=dev), generator=g)
out = torch.empty(shape, dtype=torch.float32, device=dev)
rng = np.random.default_rng(0)
for (pz, py, px) in ((3, 3, 3), (5, 5, 5), (7, 5, 7), (9, 7, 7), (9, 9, 9), (9, 3, 9), (11, 7, 7), (11, 9, 9)):
    kzx = np.abs(rng.normal(1.0, 0.5, (pz, px))) + 0.05
    ky = np.abs(rng.normal(1.0, 0.4, py)) + 0.05
    psf = (ky[None, :, None] * kzx[:, None, :]).astype(np.float32); psf /= psf.sum()
    row = {"psf": [pz, py, px], "grid": list(shape)}
    for name, fused in (("one_launch_ms", "always"), ("two_launch_ms", "never")):
        plan = RichardsonLucyPlan(shape, psf, dev, fused=fused)
        ypad = plan.new_padded_input(); ypad.view.copy_(y)
        plan(ypad, iterations=1, out=out); torch.cuda.synchronize()
        best = None
        for _ in range(3):
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            plan(ypad, iterations=4, out=out, events=ev); torch.cuda.synchronize()
            ms = ev[0].elapsed_time(ev[1]) / 4
            best = ms if best is None else min(best, ms)
        row[name] = best; row[name.replace("_ms", "_path")] = plan.path
        plan.release(); del plan, ypad
        torch.cuda.empty_cache()
    print(json.dumps(row), flush=True)
