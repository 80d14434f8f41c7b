import sys; sys.path.insert(0, '.')
import numpy as np, torch
from shrimpy_amd import _lib
from shrimpy_amd.deconvolve import RichardsonLucyPlan
dev = torch.device("cuda:0")
pshape, vshape = (13, 9, 9), (20, 64, 200)
rng = np.random.default_rng(3)
factors = [np.abs(rng.normal(1.0, 0.4, n)).astype(np.float32) + 0.05 for n in pshape]
factors = [f / f.sum() for f in factors]
y = torch.as_tensor((rng.random(vshape) * 80 + 1).astype(np.float32), device=dev)
plan = RichardsonLucyPlan(vshape, None, dev, psf_factors=factors, fused="never")
good = [plan(y, iterations=n).clone() for n in (1, 2, 3)]
torch.cuda.synchronize()
for mode in ("sync-before-stats", "no-sync", "plain-only-no-sync", "stats-only"):
    bad_before = bad_after = 0
    for trial in range(40):
        if mode == "stats-only":
            xs = []
        else:
            xs = [plan(y, iterations=n) for n in (1, 2, 3)]
        if mode == "sync-before-stats":
            torch.cuda.synchronize()
            bad_before += sum(int(not torch.equal(a, b)) for a, b in zip(xs, good))
        if mode != "plain-only-no-sync":
            xst = plan(y, iterations=3, stats=True)
            if not torch.equal(xst, good[2]): bad_after += 100
        torch.cuda.synchronize()
        bad_after += sum(int(not torch.equal(a, b)) for a, b in zip(xs, good))
    print(mode, "bad before stats run", bad_before, "bad after", bad_after)
