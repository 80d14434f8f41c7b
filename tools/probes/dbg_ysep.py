"""One compiled <PZ, PYX> instance of the fused ky (x) kzx kernel per process: a few volume shapes, bit equality with the
two-launch form.  python tools/probes/dbg_ysep.py PZ PYX"""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from shrimpy_amd.deconvolve import RichardsonLucyPlan
pz, pyx = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
rng = np.random.default_rng(100 * pz + pyx)
ok = True
for shape, iters in (((27, 9, 55), 3), ((7, 6, 59), 2), ((16, 54, 42), 1), ((5, 70, 300), 2), ((1, 1, 1), 1), ((30, 33, 129), 3)):
    for (py, px) in ((pyx, max(pyx - 2, 1)), (max(pyx - 2, 1), pyx)):
        kzx = np.abs(rng.normal(1.0, 0.5, (pz, px))) + 0.05
        ky = np.abs(rng.normal(1.0, 0.4, py)) + 0.05
        psf = (ky[None, :, None] * kzx[:, None, :]).astype(np.float32); psf /= psf.sum()
        y = torch.as_tensor((rng.random(shape) * 80 + 1).astype(np.float32), device=dev)
        plan = RichardsonLucyPlan(shape, psf, dev, fused="always")
        if plan.path != "y-separable (fused)":
            continue
        a = plan(y, iterations=iters); torch.cuda.synchronize()
        b = RichardsonLucyPlan(shape, psf, dev)(y, iterations=iters); torch.cuda.synchronize()
        eq = bool(torch.equal(a, b))
        ok &= eq
        if not eq: print("MISMATCH", (pz, py, px), shape, iters, flush=True)
print("instance", (pz, pyx), "ok" if ok else "BAD", flush=True)
sys.exit(0 if ok else 1)
