"""Benchmark of the hot path: deskew + 20-iteration Richardson-Lucy on a synthetic raw stack.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
For N > 1 it is launched as ``python -m torch.distributed.run --nproc-per-node N ... bench.py
--gpus N ...``: one process per GPU; positions are independent units, so ranks share nothing on
the data path (weak scaling, no collective) -- RCCL carries only the timing barrier / max.

One "step" = one raw volume through deskew (fused averaging) + 20 RL iterations, input already
resident in HBM.  Workload at N = 1: BASELINE.json configs[1], "2048x2048x512 f32", mapped to raw
(Z_scan=2048, Y_tilt=512, X=2048) (SURVEY.md section 8 preamble).  ``value`` = input voxels / s
over all ranks.

Extra objects on the JSON line:
  roofline     -- the dominant kernel (one fused RL iteration per launch; with --rl two-launch the
                  ratio / update launch): algorithmic 12 B/voxel (x, y in, x out; resp. in + aux +
                  out) x N_o voxels / its average launch duration, measured with HIP events on
                  the launch stream inside the timed steps, vs the 8 TB/s HBM peak.  SURVEY 8(d)
                  prices an RL iteration at 24 B/voxel (two kernels); that accounting of the same
                  launch is added as roofline.survey_8d_iteration.
  cpu_baseline -- oracle/cpu_ref.py (scipy.ndimage port of the same path) timed on the host
                  cores of this box on a bounded sample; rank 0 at N = 1 only.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

METRIC = "voxels/sec deskew+20-iter RL deconv, 2048×2048×512 f32; HBM GB/s vs peak"
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 (vector)

WORKLOADS = {
    # name: raw (Z_scan, Y_tilt, X)
    "config2": (2048, 512, 2048),   # BASELINE configs[1]: 2048x2048x512 f32, the headline
    "config4": (2048, 256, 2048),   # one position of the 96-position plate
    "config1": (256, 64, 256),      # the CPU-runnable plumbing case
    "small": (512, 128, 512),
}
DESKEW = dict(ls_angle_deg=30.0, px_to_scan_ratio=0.755, keep_overhang=False, average_n_slices=3)
PSF_SHAPE, PSF_SIGMA = (9, 7, 7), (2.0, 1.2, 1.2)
RL_ITERS = 20


def gaussian_factors():
    import numpy as np

    ks = []
    for n, s in zip(PSF_SHAPE, PSF_SIGMA):
        g = np.exp(-0.5 * ((np.arange(n) - n // 2) / s) ** 2)
        ks.append((g / g.sum()).astype(np.float32))
    return ks


def rotated_psf():
    """Secondary, non-separable PSF: the Gaussian rotated 30 deg about Y."""
    import math

    import numpy as np

    cz, cy, cx = (n // 2 for n in PSF_SHAPE)
    z, y, x = np.meshgrid(np.arange(PSF_SHAPE[0]) - cz, np.arange(PSF_SHAPE[1]) - cy,
                          np.arange(PSF_SHAPE[2]) - cx, indexing="ij")
    a = math.radians(30.0)
    zr, xr = math.cos(a) * z + math.sin(a) * x, -math.sin(a) * z + math.cos(a) * x
    g = np.exp(-0.5 * ((zr / PSF_SIGMA[0]) ** 2 + (y / PSF_SIGMA[1]) ** 2 + (xr / PSF_SIGMA[2]) ** 2))
    return (g / g.sum()).astype(np.float32)


def synthetic_raw(shape, seed, device):
    """Bead scene: sparse beads on background 100, PSF-blurred, Poisson noise, f32."""
    import torch

    from shrimpy_amd.deconvolve import correlate3d

    g = torch.Generator(device=device).manual_seed(int(seed))
    n = shape[0] * shape[1] * shape[2]
    vol = torch.zeros(shape, dtype=torch.float32, device=device)
    k = max(1, int(round(2e-5 * n)))
    idx = torch.randint(0, n, (k,), device=device, generator=g)
    amp = torch.empty(k, device=device).uniform_(200.0, 4000.0, generator=g) * 30.0
    vol.view(-1)[idx] = amp
    vol = correlate3d(vol, weight_factors=gaussian_factors())
    vol += 100.0
    return torch.poisson(vol, generator=g)


# ------------------------------------------------------------------------------------ CPU leg


def _cpu_worker(args):
    """One independent position through the oracle (runs in a spawned process, no GPU)."""
    shape, seed = args
    os.environ["OMP_NUM_THREADS"] = "1"
    import numpy as np

    from oracle import cpu_ref as o

    psf, factors = o.gaussian_psf(PSF_SHAPE, PSF_SIGMA)
    rng = np.random.default_rng(seed)
    raw = rng.poisson(100.0, size=shape).astype(np.float32)
    t0 = time.perf_counter()
    d = o.deskew(raw, DESKEW["ls_angle_deg"], DESKEW["px_to_scan_ratio"], DESKEW["keep_overhang"],
                 DESKEW["average_n_slices"])
    x = o.richardson_lucy_separable(d, factors, iterations=RL_ITERS)
    dt = time.perf_counter() - t0
    return dt, float(x.mean())


def cpu_baseline(sample_shape=(640, 128, 640), max_procs=16):
    """Oracle (kind "port": scipy.ndimage deskew + separable correlate1d RL) on the host cores."""
    import multiprocessing as mp

    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    procs = max(1, min(max_procs, avail))
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(procs) as pool:
        res = pool.map(_cpu_worker, [(sample_shape, 9000 + i) for i in range(procs)])
    wall = time.perf_counter() - t0
    busy = max(r[0] for r in res)
    voxels = procs * sample_shape[0] * sample_shape[1] * sample_shape[2]
    return {
        "value": voxels / busy,
        "unit": "voxels/s",
        "cores": procs,
        "kind": "port",
        "sample": (f"{procs} independent raw {sample_shape[0]}x{sample_shape[1]}x{sample_shape[2]} f32 "
                   f"positions, one per process (scipy.ndimage affine_transform deskew avg3 + "
                   f"{RL_ITERS}-iter RL as separable correlate1d passes), slowest worker "
                   f"{busy:.1f}s, pool wall {wall:.1f}s"),
    }


# ------------------------------------------------------------------------------------ GPU leg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS))
    ap.add_argument("--psf", default="separable", choices=["separable", "dense", "rotated"],
                    help="separable = the declared default Gaussian (rank-1 path); dense = the "
                         "rotated non-separable PSF through the 441-tap dense stencil; rotated = the "
                         "same PSF with the plan free to split it (it separates along y: a (z, x) "
                         "stencil plus a y pass per correlation)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rl", default="fused", choices=["fused", "two-launch"],
                    help="separable PSF: one launch per RL iteration (default) or the ratio / update pair")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")

    # CPU baseline first (rank 0, N = 1 only), in spawned processes that never touch the GPU.
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    import torch
    import torch.distributed as dist

    from shrimpy_amd.deconvolve import RichardsonLucyPlan
    from shrimpy_amd.deskew import fast_deskew_zyx, get_deskewed_data_shape

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU fallback for the product path")
    n_dev = torch.cuda.device_count()
    shared = int(os.environ.get("LOCAL_WORLD_SIZE", str(world))) > n_dev   # rehearsal: ranks share a card
    torch.cuda.set_device(local_rank % n_dev)
    device = torch.device("cuda", local_rank % n_dev)
    if world > 1:
        if shared:   # RCCL refuses duplicate devices; gloo carries the barrier / max (no data-path collective)
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    raw_shape = WORKLOADS[args.workload]
    out_shape, _ = get_deskewed_data_shape(raw_shape, **DESKEW)
    n_in = raw_shape[0] * raw_shape[1] * raw_shape[2]
    n_o = out_shape[0] * out_shape[1] * out_shape[2]

    # unit = (position, timepoint); rank r owns position r (weak scaling). Seeds per SURVEY 8(d).
    config_id = {"config1": 1, "config2": 2, "config4": 4, "small": 0}[args.workload]
    raw = synthetic_raw(raw_shape, seed=1000 * config_id + 7 * rank, device=device)
    if args.psf == "separable":
        plan = RichardsonLucyPlan(out_shape, None, device, psf_factors=gaussian_factors(),
                                  fused="auto" if args.rl == "fused" else "never")
    elif args.psf == "rotated":
        plan = RichardsonLucyPlan(out_shape, rotated_psf(), device)
    else:
        plan = RichardsonLucyPlan(out_shape, rotated_psf(), device, separable="never")
    # the deskew kernel writes straight into the RL kernels' padded, line-aligned input volume
    deskewed = plan.new_padded_input()
    estimate = torch.empty(out_shape, dtype=torch.float32, device=device)

    from shrimpy_amd.deskew import deskew_with_matrix
    from shrimpy_amd.geometry import deskew_geometry

    geo = deskew_geometry(raw_shape, **DESKEW)

    def step(ev=None):
        if ev:
            ev[0].record()
        deskew_with_matrix(raw, geo.matrix_3x4, geo.pre_average_shape, DESKEW["average_n_slices"],
                           out=deskewed)
        if ev:
            ev[1].record()
        plan(deskewed, iterations=RL_ITERS, out=estimate, events=(ev[3], ev[4]) if ev else None)
        if ev:
            ev[2].record()

    # sanity: the drop-in entry point gives the same tensor as the preallocated-output form
    if rank == 0 and args.workload in ("config1", "small"):
        padded = deskew_with_matrix(raw, geo.matrix_3x4, geo.pre_average_shape, 3, out=plan.new_padded_input())
        assert torch.equal(fast_deskew_zyx(raw_data=raw, **DESKEW), padded.view)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(events[i])
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if (world > 1 and shared) else device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    deskew_ms = sum(e[0].elapsed_time(e[1]) for e in events) / args.steps
    rl_ms = sum(e[1].elapsed_time(e[2]) for e in events) / args.steps          # incl. x0 = y copy
    rl_kernels_ms = sum(e[3].elapsed_time(e[4]) for e in events) / args.steps  # the 2*iters launches
    assert torch.isfinite(estimate).all(), "non-finite RL output"

    if rank == 0:
        fused = bool(getattr(plan, "fused", False))
        ysep = plan.path.startswith("y-separable")
        launches = RL_ITERS if fused else (4 * RL_ITERS if plan.path.endswith("(4 launches)") else 2 * RL_ITERS)
        launch_s = rl_kernels_ms * 1e-3 / launches  # HIP events right around the launches, / count
        # fused iteration: x, y in, x_new out; ratio / update launch: in + aux + out
        bytes_per_launch = 12.0 * n_o
        achieved = bytes_per_launch / launch_s / 1e9
        # deskew 4 N_in + 4 N_o; RL 24 B/voxel/iteration (the two-kernel accounting:
        # x, y -> ratio; ratio, x -> x) + 8 N_o init/copy.  The fused iteration's own minimum is
        # 12 B/voxel/iteration; both are reported.
        total_bytes = 4.0 * n_in + 4.0 * n_o + 24.0 * RL_ITERS * n_o + 8.0 * n_o
        min_bytes = 4.0 * n_in + 4.0 * n_o + 12.0 * launches * n_o
        ms_per_step = elapsed / args.steps * 1e3
        traffic = None
        tfile = ROOT / "profiles" / "traffic.json"
        if tfile.exists():
            try:
                rec = json.loads(tfile.read_text())
                key = "fused" if fused else ("two-launch" if args.psf == "separable" else args.psf)
                rec = rec.get(key, {})
                if rec.get("workload") == args.workload:
                    traffic = rec.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": METRIC,
            "value": world * args.steps * n_in / elapsed,
            "unit": "voxels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": (f"{args.workload}: raw (Z_scan,Y_tilt,X)={raw_shape} f32 -> deskew 30deg "
                             f"r=0.755 no-overhang avg3 -> {tuple(out_shape)} -> {RL_ITERS}-iter RL, "
                             f"{args.psf} 9x7x7 PSF; one position per GPU"),
                "raw_shape": list(raw_shape),
                "deskewed_shape": list(out_shape),
                "psf": args.psf,
                "rl_path": plan.path,
                "rl_iterations": RL_ITERS,
                "deskew_ms": deskew_ms,
                "rl_ms": rl_ms,
                "rl_kernels_ms": rl_kernels_ms,
                "rl_launches": launches,
                "algorithmic_bytes_per_step": total_bytes,
                "whole_step_hbm_frac": total_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "launched_kernels_min_bytes_per_step": min_bytes,
                "launched_kernels_hbm_frac": min_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "parallelism": f"positions x{world} (independent units, no data-path collective)"
                               + (" -- REHEARSAL: ranks share one GPU, gloo barrier" if (world > 1 and shared) else ""),
            },
            "roofline": (
                {
                    "kernel": ("rl_fused_sep_kernel<9,7> (one RL iteration per launch)" if fused
                               else "correlate_sep_kernel<9,7,7> (RL ratio / update launch)"),
                    "bound": "hbm",
                    "achieved": achieved,
                    "peak": HBM_PEAK_GBS,
                    "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS,
                    "traffic": traffic,
                    "launch_ms": launch_s * 1e3,
                    "algorithmic_bytes_per_launch": bytes_per_launch,
                    # one launch = one RL iteration, which SURVEY 8(d) prices at 24 B/voxel (two
                    # kernels, ratio through HBM); `achieved` above uses the fused kernel's own 12
                    **({"survey_8d_iteration": {"bytes": 24.0 * n_o,
                                                "achieved": 24.0 * n_o / launch_s / 1e9,
                                                "frac": 24.0 * n_o / launch_s / 1e9 / HBM_PEAK_GBS}}
                       if fused else {}),
                } if args.psf == "separable" else ({
                    # one launch per correlation: y pass (7 taps) + (z, x) stencil (63 taps) on the
                    # staged plane, in + aux + out = 12 B/voxel
                    "kernel": "correlate_dense_kernel<9,7,*,2> (ky (x) kzx, RL ratio / update launch)",
                    "bound": "hbm",
                    "achieved": achieved,
                    "peak": HBM_PEAK_GBS,
                    "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS,
                    "traffic": traffic,
                    "launch_ms": launch_s * 1e3,
                    "algorithmic_bytes_per_launch": bytes_per_launch,
                } if ysep else {
                    # a 441-tap dense stencil is fp32-VALU-bound, not HBM-bound
                    "kernel": "correlate_dense_kernel<9,7> (dense RL ratio / update launch)",
                    "bound": "valu-fp32",
                    "achieved": 2.0 * 441 * n_o / launch_s / 1e12,
                    "peak": FP32_VALU_PEAK_TFLOPS,
                    "unit": "TFLOP/s",
                    "frac": 2.0 * 441 * n_o / launch_s / 1e12 / FP32_VALU_PEAK_TFLOPS,
                    "traffic": traffic,
                    "launch_ms": launch_s * 1e3,
                    "algorithmic_flop_per_launch": 2.0 * 441 * n_o,
                    "hbm_algorithmic_GBps": achieved,
                })
            ),
        }
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
