"""Units sharded over real GPUs, results gathered to rank 0 (RCCL point-to-point), checked against a
single-rank run of the same units.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port 29513 tools/sharded_gather_check.py [--units 7] [--shape 96,24,70]

One rank per GPU (``nccl`` = RCCL; ``--backend gloo``: a rehearsal with the ranks sharing one card).  Every rank reconstructs its round-robin share (deskew + RL,
real kernels) with ``pipeline.run_sharded``; ``gather_to_rank0`` then sends the device tensors to
rank 0, which recomputes every unit itself and compares bit for bit.  Prints one JSON line.
"""

from __future__ import annotations

import argparse
import json
import os
import sys

from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--units", type=int, default=7)
    ap.add_argument("--shape", default="96,24,70")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: the ranks may share one GPU (results staged through the host)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    from shrimpy_amd.pipeline import VolumeReconstructor, gather_to_rank0, run_sharded
    from shrimpy_amd.settings import DeconvolveSettings, DeskewSettings, ReconstructSettings

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.backend == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit(f"{world} ranks need {world} GPUs, {torch.cuda.device_count()} visible")
    dev = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist.init_process_group(args.backend, device_id=dev if args.backend == "nccl" else None)

    shape = tuple(int(v) for v in args.shape.split(","))
    settings = ReconstructSettings(
        deskew=DeskewSettings(pixel_size_um=0.1133, scan_step_um=0.15, ls_angle_deg=30.0, average_n_slices=3),
        deconvolution=DeconvolveSettings(iterations=args.iters, gaussian_shape_zyx=(5, 5, 5),
                                         gaussian_sigma_zyx=(1.2, 1.0, 1.0)))
    rec = VolumeReconstructor(shape, settings, dev)

    def load(u):
        return np.random.default_rng(100 + u).integers(90, 900, shape).astype(np.uint16)

    mine = {}
    rep = run_sharded(list(range(args.units)), load, lambda v: rec(v).clone(),
                      lambda u, out: mine.__setitem__(u, out), synchronize=torch.cuda.synchronize)
    gathered = gather_to_rank0([mine[u] for u in rep.units], args.units)
    ok = True
    if rank == 0:
        assert len(gathered) == args.units and all(t.device == dev for t in gathered)
        for u in range(args.units):
            ok &= bool(torch.equal(gathered[u], rec(load(u))))
        print(json.dumps({"check": "sharded_gather", "backend": dist.get_backend(), "world_size": world,
                          "units": args.units, "units_per_rank": [len(range(r, args.units, world)) for r in range(world)],
                          "equal_to_single_rank": ok}))
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
