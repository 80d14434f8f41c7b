"""A few Richardson-Lucy iterations on a synthetic volume, for timing and for rocprofv3 passes:

    python3 tools/run_rl.py --psf rotated --rl fused --iters 4 [--grid 171,2048,2270]

Prints one JSON line: plan path, ms per launch (HIP events around the launches), algorithmic GB/s."""
import argparse
import json
import sys

from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--psf", default="rotated", choices=["rotated", "separable", "dense"])
    ap.add_argument("--rl", default="fused", choices=["fused", "two-launch"])
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--grid", default="171,2048,2270")
    ap.add_argument("--stats", action="store_true", help="sum the iteration scalars in the epilogues (the *_stats entries)")
    args = ap.parse_args()
    import torch

    import bench
    from shrimpy_amd.deconvolve import RichardsonLucyPlan

    dev = torch.device("cuda:0")
    shape = tuple(int(v) for v in args.grid.split(","))
    g = torch.Generator(device=dev).manual_seed(5)
    y = torch.poisson(torch.full(shape, 100.0, device=dev), generator=g)
    fused = "auto" if args.rl == "fused" else "never"
    if args.psf == "separable":
        plan = RichardsonLucyPlan(shape, None, dev, psf_factors=bench.gaussian_factors(), fused=fused)
    elif args.psf == "rotated":
        plan = RichardsonLucyPlan(shape, bench.rotated_psf(), dev, fused=fused)
    else:
        plan = RichardsonLucyPlan(shape, bench.rotated_psf(), dev, separable="never")
    ypad = plan.new_padded_input()
    ypad.view.copy_(y)
    out = torch.empty(shape, dtype=torch.float32, device=dev)
    plan(ypad, iterations=1, out=out)
    torch.cuda.synchronize()
    best = None
    for _ in range(args.reps):
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        plan(ypad, iterations=args.iters, out=out, events=ev, stats=args.stats)
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1])
        best = ms if best is None else min(best, ms)
    launches = {"fused": 1, "y-separable (fused)": 1, "y-separable (4 launches)": 4}.get(plan.path, 2) * args.iters
    n = shape[0] * shape[1] * shape[2]
    print(json.dumps({"path": plan.path, "grid": list(shape), "iters": args.iters, "ms_per_launch": best / launches,
                      "ms_per_iteration": best / args.iters, "algorithmic_GBps_per_launch": 12.0 * n / (best / launches) / 1e6,
                      "stats": args.stats, "flux_over_sum_y": (None if not args.stats else
                                                               [float(v) / float(y.double().sum()) for v in plan.last_stats.flux]),
                      "checksum": float(out.double().mean())}))


if __name__ == "__main__":
    main()
