import sys; sys.path.insert(0, '.')
import numpy as np, torch
from shrimpy_amd.deconvolve import RichardsonLucyPlan
want = int(sys.argv[1])
dev = torch.device("cuda:0")
rng = np.random.default_rng(45)
for case in range(10):
    pz, py, px = int(rng.choice([3, 5, 7, 9, 11])), int(rng.choice([3, 5, 9, 13, 15])), int(rng.choice([3, 5, 7, 9]))
    kzx = np.abs(rng.normal(1.0, 0.5, (pz, px))) + 0.05
    ky = np.abs(rng.normal(1.0, 0.4, py)) + 0.05
    psf = (ky[None, :, None] * kzx[:, None, :]).astype(np.float32)
    psf /= psf.sum()
    shape = (int(rng.integers(1, 30)), int(rng.integers(1, 80)), int(rng.integers(1, 200)))
    y = (rng.random(shape) * 80 + 1).astype(np.float32)
    iters = int(rng.integers(1, 4))
    if case != want: continue
    plan = RichardsonLucyPlan(shape, psf, dev, fused="always")
    print("case", case, (pz, py, px), shape, iters, plan.path, flush=True)
    got = plan(torch.as_tensor(y, device=dev), iterations=iters); torch.cuda.synchronize()
    if py <= 9:
        two = RichardsonLucyPlan(shape, psf, dev)(torch.as_tensor(y, device=dev), iterations=iters)
        print("   equal", bool(torch.equal(got, two)), flush=True)
