"""End-to-end check of the row-slab split under torch.distributed (one volume over several ranks).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port 29511 tools/slab_check.py [--backend nccl|gloo] [--shape Z,Y,X] [--iters K]

With ``--backend gloo`` the ranks may share one GPU (halos are staged through the host); with
``nccl`` (RCCL) each rank takes GPU LOCAL_RANK and the halos go GPU to GPU.  Rank 0 also runs the
unsplit volume and prints whether the reassembled slabs equal it bit for bit.
"""

from __future__ import annotations

import argparse
import os
import sys
import time

from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="gloo", choices=["gloo", "nccl"])
    ap.add_argument("--shape", default="24,96,200")
    ap.add_argument("--iters", type=int, default=6)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    from shrimpy_amd.deconvolve import RichardsonLucyPlan
    from shrimpy_amd.pipeline import gaussian_psf_factors
    from shrimpy_amd.slab import SlabRichardsonLucy

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local if args.backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(args.backend, device_id=dev if args.backend == "nccl" else None)

    shape = tuple(int(v) for v in args.shape.split(","))
    factors = gaussian_psf_factors((9, 7, 7), (2.0, 1.2, 1.2))
    g = torch.Generator(device="cpu").manual_seed(17)          # the same volume on every rank
    y = (torch.rand(shape, generator=g) * 200 + 20).to(dev)

    srl = SlabRichardsonLucy(shape, factors, dev, rank, world)
    s = srl.slab
    srl.y_pad.view.copy_(y[:, s.ext0:s.ext1, :])
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    part = srl.run(iterations=args.iters).contiguous()
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0

    parts = [None] * world
    dist.all_gather_object(parts, part.cpu().numpy())
    if rank == 0:
        whole = RichardsonLucyPlan(shape, None, dev, psf_factors=factors)(y, iterations=args.iters).cpu().numpy()
        same = bool(np.array_equal(np.concatenate(parts, axis=1), whole))
        print(f"slab split over {world} ranks ({args.backend}): equal to the unsplit run: {same}; "
              f"{args.iters} iterations in {dt * 1e3:.1f} ms")
        if not same:
            raise SystemExit(1)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
