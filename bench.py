"""Benchmark of the hot path: deskew + 20-iteration Richardson-Lucy on a synthetic raw stack.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
For N > 1 it is launched as ``python -m torch.distributed.run --nproc-per-node N ... bench.py
--gpus N ...``: one process per GPU; positions are independent units, so ranks share nothing on
the data path (weak scaling, no collective) -- RCCL carries only the timing barrier / max.

One "step" = one raw volume through deskew (fused averaging) + 20 RL iterations, input already
resident in HBM.  Workload at N = 1: BASELINE.json configs[1], "2048x2048x512 f32", mapped to raw
(Z_scan=2048, Y_tilt=512, X=2048) (SURVEY.md section 8 preamble).  ``value`` = input voxels / s
over all ranks.

``--workload config4 | config5`` run the plate / time-lapse configurations at their real per-unit
size and print the same schema: ``value`` is again the kernels-resident rate of one unit per GPU
(config 4: uint16 raw (2048, 256, 2048), deskew + RL; config 5: uint16 raw (2048, 200, 2048),
deskew -> affine registration -> RL), and ``config.store_to_store`` holds the rate of the same units
going OME-Zarr store -> ``cli.run_store`` (pinned staging, copy streams) -> OME-Zarr store.

Extra objects on the JSON line:
  roofline     -- the dominant kernel (one fused RL iteration per launch; with --rl two-launch the
                  ratio / update launch): algorithmic 12 B/voxel (x, y in, x out; resp. in + aux +
                  out) x N_o voxels / its average launch duration, measured with HIP events on
                  the launch stream inside the timed steps, vs the 8 TB/s HBM peak.
                  ``traffic`` = HBM bytes per launch from the PMC counters (profiles/traffic.json),
                  reported only while that record's kernel-source fingerprint equals this
                  checkout's (``_lib.kernel_source_sha16``) -- null otherwise.
  cpu_baseline -- oracle/cpu_ref.py (scipy.ndimage port of the same path) timed on the host
                  cores of this box on a bounded sample; rank 0 at N = 1 only (BASELINE.md section 4:
                  all host cores on independent positions, plus a single-core leg labelled
                  extrapolated).
"""

from __future__ import annotations

import argparse
import json
import math
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
    # name: (config id for the seeds, raw (Z_scan, Y_tilt, X))
    "config2": (2, (2048, 512, 2048)),   # BASELINE configs[1]: 2048x2048x512 f32, the headline
    "config4": (4, (2048, 256, 2048)),   # one position of the 96-position plate (uint16 counts)
    "config5": (5, (2048, 200, 2048)),   # one (t, p) unit of the time-lapse: deskew -> register -> RL
    "config1": (1, (256, 64, 256)),      # the CPU-runnable plumbing case
    "small": (0, (512, 128, 512)),
}
PLATE_WORKLOADS = ("config4", "config5")
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
    import numpy as np

    cz, cy, cx = (n // 2 for n in PSF_SHAPE)
    z, y, x = np.meshgrid(np.arange(PSF_SHAPE[0]) - cz, np.arange(PSF_SHAPE[1]) - cy,
                          np.arange(PSF_SHAPE[2]) - cx, indexing="ij")
    a = math.radians(30.0)
    zr, xr = math.cos(a) * z + math.sin(a) * x, -math.sin(a) * z + math.cos(a) * x
    g = np.exp(-0.5 * ((zr / PSF_SIGMA[0]) ** 2 + (y / PSF_SIGMA[1]) ** 2 + (xr / PSF_SIGMA[2]) ** 2))
    return (g / g.sum()).astype(np.float32)


def measured_psf(shape=(15, 19, 19)):
    """A dense, non-separable PSF of the size of a measured bead patch (15 x 18 x 18 voxels, padded to odd:
    /root/reference/scripts/measure_psf.py:187-190 at run time of the survey): a tilted Gaussian with a weak off-axis
    lobe.  Runs the RL iteration in the Fourier domain (shrimpy_amd/deconvolve_fft.py)."""
    import numpy as np

    z, y, x = np.meshgrid(*[np.arange(n) - n // 2 for n in shape], indexing="ij")
    a = math.radians(30.0)
    zr, xr = math.cos(a) * z + math.sin(a) * x, -math.sin(a) * z + math.cos(a) * x
    g = np.exp(-0.5 * ((zr / 3.0) ** 2 + (y / 2.2) ** 2 + (xr / 2.2) ** 2))
    g += 0.04 * np.exp(-0.5 * (((z - 2) / 2.5) ** 2 + ((y + 3) / 2.0) ** 2 + ((x - 3) / 1.8) ** 2))
    return (g / g.sum()).astype(np.float32)


def registration_matrix():
    """rotation 2 deg about Z, scale (1, .98, 1.02), translation (3.5, -12.25, 20.75)."""
    import numpy as np

    th = np.deg2rad(2.0)
    rot = np.array([[1, 0, 0], [0, np.cos(th), -np.sin(th)], [0, np.sin(th), np.cos(th)]])
    m = np.eye(4)
    m[:3, :3] = rot @ np.diag([1.0, 0.98, 1.02])
    m[:3, 3] = [3.5, -12.25, 20.75]
    return m


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
    """One independent position through the oracle (runs in a spawned process, no GPU): the bead
    scene of SURVEY 8(d) with that position's seed, then deskew + RL, timed."""
    shape, seed = args
    os.environ["OMP_NUM_THREADS"] = "1"

    from oracle import cpu_ref as o

    _, factors = o.gaussian_psf(PSF_SHAPE, PSF_SIGMA)
    raw = o.bead_scene(shape, seed, psf_factors=factors)
    t0 = time.perf_counter()
    d = o.deskew(raw, DESKEW["ls_angle_deg"], DESKEW["px_to_scan_ratio"], DESKEW["keep_overhang"],
                 DESKEW["average_n_slices"])
    x = o.richardson_lucy_separable(d, factors, iterations=RL_ITERS)
    dt = time.perf_counter() - t0
    return dt, float(x.mean())


def host_cores() -> int:
    """Cores this process may use (affinity mask cut down to a cgroup CPU quota)."""
    from shrimpy_amd.io.omezarr import host_cores as cores

    return cores()


def cpu_baseline(config_id=2, all_core_shape=(640, 128, 640), single_shape=(512, 128, 512), max_procs=None):
    """Oracle (kind "port": scipy.ndimage deskew + separable correlate1d RL) on the host cores,
    per BASELINE.md section 4: (ii) every host core runs one independent position (aggregate rate,
    core count stated); (i) one core runs the reduced (512, 128, 512) volume, three samples, median,
    scaled linearly in voxels to config 2 and labelled extrapolated."""
    import multiprocessing as mp
    import statistics

    cores = host_cores()
    # every host core, as BASELINE.md section 4 says -- up to 64 workers (about 1 GB each while the scene is
    # generated: a whole 8-GPU node's cores at once would be a three-digit GB sample, not a bounded one)
    procs = min(cores, 64) if max_procs is None else max(1, min(cores, int(max_procs)))
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(procs) as pool:
        res = pool.map(_cpu_worker, [(all_core_shape, 1000 * config_id + 7 * i) for i in range(procs)])
    wall = time.perf_counter() - t0
    busy = max(r[0] for r in res)
    voxels = procs * all_core_shape[0] * all_core_shape[1] * all_core_shape[2]
    with ctx.Pool(min(3, cores)) as pool:   # three single-core samples side by side on an idle machine
        single = pool.map(_cpu_worker, [(single_shape, 1000 * config_id + 7 * i) for i in range(3)])
    s_med = statistics.median(r[0] for r in single)
    n_single = single_shape[0] * single_shape[1] * single_shape[2]
    n_cfg2 = 2048 * 512 * 2048
    shp = "x".join(str(v) for v in all_core_shape)
    return {
        "value": voxels / busy,
        "unit": "voxels/s",
        "cores": procs,
        "host_cores": cores,
        # fewer workers than cores (the 64-worker memory cap, or --cpu-procs): the aggregate is then a LOWER bound of
        # what the whole host would do, and any GPU/CPU ratio taken from it an upper bound
        "capped": procs < cores,
        "kind": "port",
        "sample": (f"{procs} independent raw {shp} f32 bead-scene positions (seeds 1000*{config_id}+7*p), one per "
                   f"process on {procs} of {cores} host cores (scipy.ndimage affine_transform deskew avg3 + {RL_ITERS}-iter RL "
                   f"as separable correlate1d passes), slowest worker {busy:.1f}s, pool wall {wall:.1f}s"),
        "single_core": {
            "value": n_single / s_med,
            "unit": "voxels/s",
            "cores": 1,
            "sample": f"raw {'x'.join(str(v) for v in single_shape)}, median of 3: {s_med:.1f}s",
            "config2_seconds_extrapolated": s_med * n_cfg2 / n_single,
            "label": "extrapolated (linear in voxels)",
        },
    }


# ------------------------------------------------------------------------------------ helpers


def pmc_traffic(key, workload, kernel_symbol):
    """HBM bytes per launch from the committed PMC record -- only if that record was taken from THIS
    build of the kernels (same source fingerprint, same kernel symbol, same workload)."""
    from shrimpy_amd._lib import kernel_source_sha16

    tfile = ROOT / "profiles" / "traffic.json"
    if not tfile.exists():
        return None, "no profiles/traffic.json"
    try:
        rec = json.loads(tfile.read_text()).get(key, {})
    except Exception as exc:  # noqa: BLE001
        return None, f"unreadable traffic.json: {exc}"
    if rec.get("workload") != workload:
        return None, f"record is for workload {rec.get('workload')!r}"
    have = kernel_source_sha16()
    if rec.get("source_sha16") != have:
        return None, f"stale: record {rec.get('source_sha16')} != build {have}"
    if kernel_symbol and kernel_symbol not in (rec.get("kernel") or ""):
        return None, f"record is for kernel {rec.get('kernel')!r}"
    return rec.get("hbm_bytes_per_launch"), f"{rec.get('source')} @ {rec.get('git_head', '?')}"


def library_stamp() -> dict:
    """Which binary produced the numbers: the stamp compiled into the loaded liblsrecon.so (``lsr_source_sha16``) and
    the fingerprint of the sources beside it -- equal, or ``_lib.load()`` would have refused the library."""
    from shrimpy_amd import _lib

    return {"path": str(_lib.LIB_PATH.relative_to(ROOT)) if _lib.LIB_PATH.is_relative_to(ROOT) else str(_lib.LIB_PATH),
            "source_sha16": _lib.library_source_sha16(), "checkout_sha16": _lib.kernel_source_sha16()}


def device_state(device) -> dict:
    """Clocks and power of the card under ``device``, read from sysfs (no process is started): the
    box-to-box spread of the launch time is only attributable with these next to it.  Every field is
    optional -- a box that hides sysfs gives {}."""
    import torch

    out = {}
    try:
        prop = torch.cuda.get_device_properties(device)
        out["name"] = prop.name
        out["compute_units"] = int(prop.multi_processor_count)
        bdf = None
        if all(hasattr(prop, a) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
            bdf = f"{prop.pci_domain_id:04x}:{prop.pci_bus_id:02x}:{prop.pci_device_id:02x}.0"
        cards = sorted(Path("/sys/class/drm").glob("card[0-9]*/device"))
        card = None
        for c in cards:
            try:
                if bdf and c.resolve().name.lower() == bdf:
                    card = c
                    break
            except OSError:
                continue
        if card is None and len(cards) == 1:
            card = cards[0]
        if card is None:
            return out
        out["pci"] = card.resolve().name

        def current(name):   # the starred level of a pp_dpm_* table, MHz
            try:
                for line in (card / name).read_text().splitlines():
                    if line.rstrip().endswith("*"):
                        return int("".join(ch for ch in line.split(":")[1] if ch.isdigit()))
            except (OSError, ValueError, IndexError):
                pass
            return None

        def top(name):
            try:
                vals = [int("".join(ch for ch in ln.split(":")[1] if ch.isdigit()))
                        for ln in (card / name).read_text().splitlines() if ":" in ln]
                return max(vals) if vals else None
            except (OSError, ValueError, IndexError):
                return None

        out["sclk_mhz"], out["sclk_max_mhz"] = current("pp_dpm_sclk"), top("pp_dpm_sclk")
        out["mclk_mhz"], out["mclk_max_mhz"] = current("pp_dpm_mclk"), top("pp_dpm_mclk")
        for hw in sorted(card.glob("hwmon/hwmon*")):
            for key, fname, scale in (("power_cap_w", "power1_cap", 1e-6), ("power_avg_w", "power1_average", 1e-6),
                                      ("power_input_w", "power1_input", 1e-6), ("temp_c", "temp1_input", 1e-3)):
                try:
                    out[key] = round(int((hw / fname).read_text()) * scale, 1)
                except (OSError, ValueError):
                    pass
        try:
            out["perf_level"] = (card / "power_dpm_force_performance_level").read_text().strip()
        except OSError:
            pass
    except Exception as exc:  # noqa: BLE001 -- never let bookkeeping fail a bench
        out["error"] = f"{type(exc).__name__}: {exc}"
    return {k: v for k, v in out.items() if v is not None}


def dist_setup(args):
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU fallback for the product path")
    n_dev = torch.cuda.device_count()
    shared = int(os.environ.get("LOCAL_WORLD_SIZE", str(world))) > n_dev   # rehearsal: ranks share a card
    torch.cuda.set_device(local_rank % n_dev)
    device = torch.device("cuda", local_rank % n_dev)
    backend = None
    if world > 1:
        if shared:   # RCCL refuses duplicate devices; gloo carries the barrier / max (no data-path collective)
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
        backend = dist.get_backend()
    return rank, world, device, shared, backend


def self_launch(n: int) -> int:
    """``python bench.py --gpus N`` from a bare shell: start ``torch.distributed.run`` with N ranks as a
    CHILD process (nothing here has touched the GPU yet, and a GPU-initialised process must never be
    replaced by exec on this pool), pass its output through and return its exit code."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    # the contract is ONE JSON line on stdout: whatever else the ranks print there (gloo announces its
    # connections on stdout) goes to stderr
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in (proc.stdout or "").splitlines():
        (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")
    return proc.returncode


def timed_steps(step, args, world, shared, device):
    """W untimed warm-ups, then exactly K steps between barrier + synchronize, max over ranks."""
    import torch
    import torch.distributed as dist

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
    return float(t.item()), events


def parallelism_note(world, shared):
    return (f"positions x{world} (independent units, no data-path collective)"
            + (" -- REHEARSAL: ranks share one GPU, gloo barrier" if (world > 1 and shared) else ""))


# ------------------------------------------------------------------------------------ config 2 (headline)


def run_resident(args, rank, world, device, shared, backend, cpu):
    import torch

    from shrimpy_amd.deconvolve import RichardsonLucyPlan
    from shrimpy_amd.deskew import deskew_with_matrix, fast_deskew_zyx, get_deskewed_data_shape
    from shrimpy_amd.geometry import deskew_geometry

    config_id, raw_shape = WORKLOADS[args.workload]
    out_shape, _ = get_deskewed_data_shape(raw_shape, **DESKEW)
    n_in = raw_shape[0] * raw_shape[1] * raw_shape[2]
    n_o = out_shape[0] * out_shape[1] * out_shape[2]

    # unit = (position, timepoint); rank r owns position r (weak scaling). Seeds per SURVEY 8(d).
    raw = synthetic_raw(raw_shape, seed=1000 * config_id + 7 * rank, device=device)
    if args.psf == "separable":
        plan = RichardsonLucyPlan(out_shape, None, device, psf_factors=gaussian_factors(),
                                  fused="auto" if args.rl == "fused" else "never")
    elif args.psf == "rotated":
        plan = RichardsonLucyPlan(out_shape, rotated_psf(), device, fused="never" if args.rl == "two-launch" else "auto")
    elif args.psf == "measured":
        from shrimpy_amd.deconvolve import make_plan

        plan = make_plan(out_shape, measured_psf(), device)
    else:
        plan = RichardsonLucyPlan(out_shape, rotated_psf(), device, separable="never")
    # the deskew kernel writes straight into the RL kernels' padded, line-aligned input volume (the Fourier-domain
    # plan takes a plain dense one)
    deskewed = plan.new_padded_input() if plan.padded_input else torch.empty(out_shape, dtype=torch.float32, device=device)
    estimate = torch.empty(out_shape, dtype=torch.float32, device=device)
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

    elapsed, events = timed_steps(step, args, world, shared, device)
    state_after = device_state(device) if rank == 0 else None   # right behind the timed steps: clocks under load
    deskew_ms = sum(e[0].elapsed_time(e[1]) for e in events) / args.steps
    rl_ms = sum(e[1].elapsed_time(e[2]) for e in events) / args.steps          # incl. x0 = y copy
    rl_kernels_ms = sum(e[3].elapsed_time(e[4]) for e in events) / args.steps  # the 2*iters launches
    assert torch.isfinite(estimate).all(), "non-finite RL output"
    if rank != 0:
        return None

    fused = bool(getattr(plan, "fused", False)) or bool(getattr(plan, "fused_ysep", False))
    ysep = plan.path.startswith("y-separable")
    launches = RL_ITERS if fused else (4 * RL_ITERS if plan.path.endswith("(4 launches)") else 2 * RL_ITERS)
    if plan.path == "fft":
        launches = RL_ITERS      # the roofline unit below is one ITERATION (ten launches: two convolutions of five)
    launch_s = rl_kernels_ms * 1e-3 / launches  # HIP events right around the launches, / count
    # fused iteration: x, y in, x_new out; ratio / update launch: in + aux + out
    bytes_per_launch = 12.0 * n_o
    achieved = bytes_per_launch / launch_s / 1e9
    # deskew 4 N_in + 4 N_o; RL 24 B/voxel/iteration (the two-kernel accounting:
    # x, y -> ratio; ratio, x -> x) + 8 N_o init/copy.  The fused iteration's own minimum is
    # 12 B/voxel/iteration; both are reported, the second is what the launched kernels can reach.
    survey_bytes = 4.0 * n_in + 4.0 * n_o + 24.0 * RL_ITERS * n_o + 8.0 * n_o
    min_bytes = 4.0 * n_in + 4.0 * n_o + 12.0 * launches * n_o
    ms_per_step = elapsed / args.steps * 1e3
    if plan.path == "fft":
        # per convolution: rows forward 4 N_o + 8 G, y 16 G, z leg 16 G + 8 G (the PSF's spectrum), y 16 G, rows
        # inverse 8 G + 8 N_o  (G = complex points of the half spectrum on the transform grid); two per iteration
        gz, gy, gx = plan.grid
        g_c = gz * gy * (gx // 2 + 1)
        bytes_per_launch = 2.0 * (12.0 * n_o + 72.0 * g_c)
        achieved = bytes_per_launch / launch_s / 1e9
        min_bytes = 4.0 * n_in + 4.0 * n_o + RL_ITERS * bytes_per_launch
        kernel, symbol = ("Fourier-domain RL iteration: 2 x (rfft_rows_kernel, rocFFT y, zcorr_kernel, rocFFT y, "
                          "irfft_rows_rl_kernel) on grid %s" % (tuple(plan.grid),)), "irfft_rows_rl_kernel"
    elif args.psf == "separable":
        kernel = ("rl_fused_sep_kernel<9,7> (one RL iteration per launch)" if fused
                  else "correlate_sep_kernel<9,7,7> (RL ratio / update launch)")
        symbol = "rl_fused_sep_kernel" if fused else "correlate_sep_kernel"
    elif ysep and fused:
        kernel, symbol = "rl_fused_ysep_kernel<9,7> (ky (x) kzx, one RL iteration per launch)", "rl_fused_ysep_kernel"
    elif ysep:
        kernel, symbol = "correlate_dense_kernel<9,7,*,2> (ky (x) kzx, RL ratio / update launch)", "correlate_dense_kernel"
    else:
        kernel, symbol = "correlate_dense_kernel<9,7> (dense RL ratio / update launch)", "correlate_dense_kernel"
    key = ("fused" if fused else "two-launch") if args.psf == "separable" else args.psf
    traffic, traffic_note = pmc_traffic(key, args.workload, symbol)
    if args.psf == "separable" or ysep or plan.path == "fft":
        roofline = {
            "kernel": kernel, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_record": traffic_note,
            "launch_ms": launch_s * 1e3, "algorithmic_bytes_per_launch": bytes_per_launch,
        }
    else:
        # a 441-tap dense stencil is fp32-VALU-bound, not HBM-bound
        flop = 2.0 * 441 * n_o
        roofline = {
            "kernel": kernel, "bound": "valu-fp32", "achieved": flop / launch_s / 1e12,
            "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flop / launch_s / 1e12 / FP32_VALU_PEAK_TFLOPS,
            "traffic": traffic, "traffic_record": traffic_note, "launch_ms": launch_s * 1e3,
            "algorithmic_flop_per_launch": flop, "hbm_algorithmic_GBps": achieved,
        }
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
                         f"{args.psf} {'x'.join(str(n) for n in (plan.psf.shape if plan.path == 'fft' else PSF_SHAPE))} PSF; one position per GPU"),
            "raw_shape": list(raw_shape),
            "deskewed_shape": list(out_shape),
            "psf": args.psf,
            "rl_path": plan.path,
            "rl_iterations": RL_ITERS,
            "deskew_ms": deskew_ms,
            "rl_ms": rl_ms,
            "rl_kernels_ms": rl_kernels_ms,
            "rl_launches": launches,
            "launched_kernels_min_bytes_per_step": min_bytes,
            "launched_kernels_hbm_frac": min_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
            # SURVEY 8(d) prices the step as if every RL iteration were two kernels (24 B/voxel);
            # the fused launch no longer moves those bytes -- kept for comparison with the survey's
            # 50 ms floor only, it is NOT an achieved-bandwidth figure
            "survey_8d_two_kernel_accounting": {
                "bytes_per_step": survey_bytes,
                "frac_if_those_bytes_moved": survey_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "parallelism": parallelism_note(world, shared),
            "collective_backend": backend,
            "device_after_timed_steps": state_after,
            "library": library_stamp(),
        },
        "roofline": roofline,
    }
    if cpu is not None:
        line["cpu_baseline"] = cpu
    return line


# ------------------------------------------------------------------------------------ configs 4 and 5


def plate_settings(workload):
    from shrimpy_amd.settings import DeconvolveSettings, DeskewSettings, ReconstructSettings, RegisterSettings

    return ReconstructSettings(
        deskew=DeskewSettings(pixel_size_um=0.1133, scan_step_um=0.15, ls_angle_deg=30.0, keep_overhang=False,
                              average_n_slices=3),
        registration=(RegisterSettings(affine_transform_zyx=registration_matrix().tolist())
                      if workload == "config5" else None),
        deconvolution=DeconvolveSettings(iterations=RL_ITERS))


def _blosc_backend() -> str:
    from shrimpy_amd.io.codecs import blosc_backend

    return blosc_backend()


def host_floor(root, key, out_shape):
    """What the host side of one unit costs at best on this rank's cores: the unit's read (file reads + entropy
    decoding + unshuffle into one buffer) and its write (chunk files of the float32 result) each timed on ONE thread,
    added and divided by the core share -- the store-to-store time per unit if both pools scaled perfectly and the
    kernels, copies and Python cost nothing.  The measured ``s_per_unit`` next to it says how far the pipeline is
    from its host."""
    import numpy as np

    from shrimpy_amd.io.omezarr import as_volume_array, io_thread_budget, open_ome_zarr, rank_cores

    def timed_io(read_threads, write_threads):
        prev = io_thread_budget(read=read_threads, write=write_threads)
        try:
            with open_ome_zarr(root / "in.zarr", mode="r", prefer_iohub=False) as plate:
                arr = as_volume_array(dict(plate.positions())[key]["0"])
                buf = np.empty(arr.shape[2:], dtype=arr.dtype)
                arr.read_volume(0, 0, out=buf)            # (page cache warm, as in the run)
                t0 = time.perf_counter()
                arr.read_volume(0, 0, out=buf)
                t_read = time.perf_counter() - t0
            with open_ome_zarr(root / "out.zarr", mode="a", prefer_iohub=False) as plate:
                arr = as_volume_array(dict(plate.positions())[key]["0"])
                vol = np.zeros(tuple(out_shape), dtype=np.float32)
                vol[::7] = 1.0
                arr.write_volume(0, 0, vol)               # (the files exist: rewrites, as every unit after the first pass of tmpfs pages)
                t0 = time.perf_counter()
                arr.write_volume(0, 0, vol)
                t_write = time.perf_counter() - t0
            return t_read, t_write, buf.nbytes, vol.nbytes
        finally:
            io_thread_budget(**prev)

    r1, w1, rbytes, wbytes = timed_io(1, 1)
    rn, wn, _, _ = timed_io(None, None)          # the run's own budgets, each stage alone on the machine
    cores = rank_cores()
    return {"read_one_thread_s": round(r1, 4), "write_one_thread_s": round(w1, 4), "cores": cores,
            "floor_s_per_unit": round((r1 + w1) / cores, 4),
            "read_GBps_per_core": round(rbytes / r1 / 1e9, 2), "write_GBps_per_core": round(wbytes / w1 / 1e9, 2),
            "read_all_threads_s": round(rn, 4), "write_all_threads_s": round(wn, 4),
            "read_GBps_all_threads": round(rbytes / rn / 1e9, 1), "write_GBps_all_threads": round(wbytes / wn / 1e9, 1),
            "floor_shared_memory_s_per_unit": round(rn + wn, 4),
            "note": "floor_s_per_unit = (one-thread read + one-thread write of one unit) / cores: perfect scaling of both "
                    "pools; floor_shared_memory_s_per_unit = the two stages with the run's thread budgets, each ALONE on "
                    "the machine, added: what a streamed run costs when reader and writer share the host's memory "
                    "system instead of overlapping (copies, the page cache and the entropy coder all move bytes through it)"}


def run_plate(args, rank, world, device, shared, backend, cpu):
    """Configs 4 / 5: (a) one unit per GPU with its uint16 stack resident in HBM through the
    production pipeline object (``VolumeReconstructor``); (b) the same kind of units store to store."""
    import shutil
    import tempfile

    import torch
    import torch.distributed as dist

    from shrimpy_amd import cli
    from shrimpy_amd.io.omezarr import open_ome_zarr
    from shrimpy_amd.pipeline import VolumeReconstructor

    config_id, raw_shape = WORKLOADS[args.workload]
    settings = plate_settings(args.workload)
    rec = VolumeReconstructor(raw_shape, settings, device)
    out_shape = rec.output_shape
    n_in = raw_shape[0] * raw_shape[1] * raw_shape[2]
    n_o = out_shape[0] * out_shape[1] * out_shape[2]
    raw16 = synthetic_raw(raw_shape, seed=1000 * config_id + 7 * rank, device=device).to(torch.uint16)
    result = {}

    def step(ev=None):
        if ev:
            ev[0].record()
        result["x"] = rec(raw16, rl_events=(ev[3], ev[4]) if ev else None)
        if ev:
            ev[2].record()

    elapsed, events = timed_steps(step, args, world, shared, device)
    step_ms = sum(e[0].elapsed_time(e[2]) for e in events) / args.steps
    rl_kernels_ms = sum(e[3].elapsed_time(e[4]) for e in events) / args.steps
    assert torch.isfinite(result["x"]).all(), "non-finite RL output"
    launch_s = rl_kernels_ms * 1e-3 / RL_ITERS
    del result

    # ---- store to store: T x P units through cli.run_store (staged path) ----
    store = None
    if not args.no_store_leg:
        # enough units that the run is its steady state, not the pipeline's fill and drain
        n_t, n_p = (1, max(12, 3 * world)) if args.workload == "config4" else (4, max(3, world))
        scratch = None
        if rank == 0:
            scratch = Path(tempfile.mkdtemp(prefix="lsr_bench_", dir=args.scratch))
            # the temporary plates (uint16 in, float32 out) must fit the scratch file system: fewer
            # positions (never fewer than one per rank) rather than a run that dies with the disk full
            per_unit = 2 * n_in + 4 * n_o
            free = shutil.disk_usage(scratch).free
            while n_p > world and 1.25 * n_t * n_p * per_unit > free:
                n_p -= 1
        root = [str(scratch), n_p]
        if world > 1:
            dist.broadcast_object_list(root, src=0)     # rank 0's directory and position count
        root, n_p = Path(root[0]), int(root[1])
        try:
            keys = [f"A/{p + 1}/0" for p in range(n_p)]
            if rank == 0:
                # the acquisition's layout: one shard per volume, blosc-zstd chunks of 32 planes, c-blosc's own 32 KB blocks
                fmt = dict(compress="blosc-zstd", shards="volume", blocksize=32768) if args.engine_format else {}
                with open_ome_zarr(root / "in.zarr", layout="hcs", mode="w", channel_names=["LS"],
                                   prefer_iohub=False, version="0.5" if args.engine_format else "0.4") as plate:
                    for p, key in enumerate(keys):
                        arr = plate.create_position(*key.split("/")).create_zeros(
                            "0", shape=(n_t, 1) + tuple(raw_shape), dtype="uint16",
                            scale=(1, 1, 0.15, 0.1133, 0.1133), **fmt)
                        for t in range(n_t):
                            v = synthetic_raw(raw_shape, seed=1000 * config_id + 7 * p + t, device=device)
                            arr.write_volume(t, 0, v.to(torch.uint16).cpu().numpy())
                            del v
            torch.cuda.empty_cache()
            if world > 1:
                dist.barrier()
            res = cli.run_store(root / "in.zarr", root / "out.zarr", settings,
                                compression=None if args.output_compression == "none" else args.output_compression,
                                zarr_version="0.5" if args.output_compression != "none" else "0.4")
            units = res["units_total"]
            store = {
                "positions": n_p, "timepoints": n_t, "units": units, "seconds": res["job_seconds"],
                "s_per_unit": res["job_seconds"] / max(units, 1) * world,
                "voxels_per_s": units * n_in / res["job_seconds"],
                # rank 0's stage clocks, seconds per unit (loader thread: wait_slot, load, stage_in; caller: wait_load,
                # process, wait_store, stage_out; writer thread: collect, write)
                "stage_s_per_unit": {k: round(v / max(res["units"], 1), 4) for k, v in res.get("stage_seconds", {}).items()},
                "device_codec": res.get("device_codec"),
                # rank 0's pace once its pipeline is full (median interval between consecutive units); `s_per_unit` above is
                # the whole job including the first read and the last write
                "steady_s_per_unit_rank0": res.get("steady_s_per_unit"),
                "io": ("native OME-Zarr reader/writer, "
                       + ("input in the acquisition's format (Zarr v3, one shard per volume, blosc-zstd chunks (1,1,32,ny,nx), "
                          f"frames decoded by {_blosc_backend()}), output uncompressed ~64 MB chunks, " if args.engine_format
                          else "uncompressed chunks (input (1,1,32,ny,nx), output ~64 MB), ")
                       + (f"OUTPUT chunks {args.output_compression}, " if args.output_compression != "none" else "")
                       + f"scratch {root.parent}, input from the page cache; pinned staging slots + copy streams (cli.run_store)"),
            }
            if world > 1:       # every rank's own stage clocks (a rehearsal on one card, or the node-wide run)
                clocks = [None] * world
                dist.all_gather_object(clocks, {"rank": rank, "units": res["units"], "seconds": round(res["seconds"], 4),
                                                **{k: round(v / max(res["units"], 1), 4) for k, v in res.get("stage_seconds", {}).items()}})
                store["stage_s_per_unit_by_rank"] = clocks
            if rank == 0:
                store["host_floor"] = host_floor(root, keys[0], out_shape)
            if world > 1:
                dist.barrier()
        finally:
            if rank == 0:
                shutil.rmtree(root, ignore_errors=True)
    if rank != 0:
        return None
    achieved = 12.0 * n_o / launch_s / 1e9
    traffic, traffic_note = pmc_traffic("fused", args.workload, "rl_fused_sep_kernel")
    chain = "deskew -> affine register (config-3 matrix) -> " if args.workload == "config5" else "deskew -> "
    line = {
        "metric": METRIC,
        "value": world * args.steps * n_in / elapsed,
        "unit": "voxels/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": (f"{args.workload}: one unit per GPU, raw (Z_scan,Y_tilt,X)={raw_shape} uint16 resident in HBM "
                         f"-> {chain}{RL_ITERS}-iter RL (separable 9x7x7 PSF) -> {tuple(out_shape)} f32, through "
                         "pipeline.VolumeReconstructor"),
            "raw_shape": list(raw_shape),
            "output_shape": list(out_shape),
            "rl_iterations": RL_ITERS,
            "kernels_resident": {"ms_per_unit": step_ms, "rl_kernels_ms": rl_kernels_ms,
                                 "voxels_per_s_per_gpu": n_in / (step_ms * 1e-3)},
            "store_to_store": store,
            "parallelism": parallelism_note(world, shared),
            "collective_backend": backend,
            "library": library_stamp(),
        },
        "roofline": {
            "kernel": "rl_fused_sep_kernel<9,7> (one RL iteration per launch)", "bound": "hbm",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_record": traffic_note, "launch_ms": launch_s * 1e3,
            "algorithmic_bytes_per_launch": 12.0 * n_o,
        },
    }
    if cpu is not None:
        line["cpu_baseline"] = cpu
    return line


def run_config1_cpu(args):
    """BASELINE configs[0]: a single 256x256x64 oblique stack (raw (256, 64, 256)), deskew only, no GPU --
    through the PRODUCT path on a CPU tensor (`fast_deskew_zyx` -> `lsr_deskew_f32_cpu`, csrc/host_twins.hip),
    with the oracle (scipy.ndimage) timed beside it on the same stack and the outputs compared bit for bit."""
    import numpy as np
    import torch

    from oracle import cpu_ref as o
    from shrimpy_amd.deskew import fast_deskew_zyx

    config_id, raw_shape = WORKLOADS["config1"]
    _, factors = o.gaussian_psf(PSF_SHAPE, PSF_SIGMA)
    raw = o.bead_scene(raw_shape, 1000 * config_id, psf_factors=factors)
    vol = torch.as_tensor(raw)
    threads = torch.get_num_threads()
    for _ in range(args.warmup):
        fast_deskew_zyx(raw_data=vol, **DESKEW)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = fast_deskew_zyx(raw_data=vol, **DESKEW)
    elapsed = time.perf_counter() - t0
    t1 = time.perf_counter()
    want = o.deskew(raw, DESKEW["ls_angle_deg"], DESKEW["px_to_scan_ratio"], DESKEW["keep_overhang"], DESKEW["average_n_slices"])
    oracle_s = time.perf_counter() - t1
    n_in = raw_shape[0] * raw_shape[1] * raw_shape[2]
    return {
        "metric": METRIC, "value": args.steps * n_in / elapsed, "unit": "voxels/s", "n_gpus": 0, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": (f"config1: raw (Z_scan,Y_tilt,X)={raw_shape} f32 -> deskew 30deg r=0.755 no-overhang avg3 -> "
                                f"{tuple(out.shape)}, deskew only, no GPU: the product's host twin on a CPU tensor"),
                   "raw_shape": list(raw_shape), "deskewed_shape": list(out.shape), "host_threads": threads,
                   "equals_oracle_bit_for_bit": bool(np.array_equal(out.numpy(), want)), "library": library_stamp()},
        "roofline": None,
        "cpu_baseline": {"value": n_in / oracle_s, "unit": "voxels/s", "cores": 1, "kind": "port",
                         "sample": f"the same stack through oracle/cpu_ref.py (scipy.ndimage.affine_transform + slice mean), once: {oracle_s:.2f}s"},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS))
    ap.add_argument("--psf", default="separable", choices=["separable", "dense", "rotated", "measured"],
                    help="separable = the declared default Gaussian (rank-1 path); dense = the "
                         "rotated non-separable PSF through the 441-tap dense stencil; rotated = the "
                         "same PSF with the plan free to split it (it separates along y: a (z, x) "
                         "stencil plus a y pass per correlation); measured = a dense 15x19x19 bead-patch PSF "
                         "(the RL iteration in the Fourier domain)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-procs", type=int, default=None, help="cap the CPU baseline's worker count (default: all host cores)")
    ap.add_argument("--rl", default="fused", choices=["fused", "two-launch", "fused-ysep"],
                    help="one launch per RL iteration (default: rl_fused_sep / rl_fused_ysep kernels) or the ratio / "
                         "update pair (two-launch); fused-ysep is accepted as a synonym of fused")
    ap.add_argument("--no-store-leg", action="store_true", help="config4/5: skip the store-to-store leg")
    ap.add_argument("--scratch", default=None, help="config4/5: directory for the temporary plates (default: TMPDIR)")
    ap.add_argument("--engine-format", action="store_true",
                    help="config4/5 store leg: input plate as the acquisition writes it (Zarr v3, one shard per "
                         "volume around blosc-zstd chunks) instead of uncompressed chunks")
    ap.add_argument("--output-compression", default="none", choices=["none", "zstd", "blosc-zstd"],
                    help="config4/5 store leg: chunk compression of the OUTPUT plate (the CLI's own default is blosc-zstd, "
                         "what the acquisition engine writes; the bench line's default leaves the float32 result raw)")
    ap.add_argument("--device", default="gpu", choices=["gpu", "cpu"],
                    help="cpu: BASELINE configs[0] (config1, deskew only) through the product's host twins, no GPU touched")
    args = ap.parse_args()

    if args.device == "cpu":
        if args.workload != "config1" or args.gpus != 1:
            raise SystemExit("--device cpu runs --workload config1 (BASELINE configs[0]: deskew only, no GPU)")
        print(json.dumps(run_config1_cpu(args)))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    # CPU baseline first (rank 0, N = 1 only), in spawned processes that never touch the GPU.
    cpu = None
    if world_env == 1 and args.gpus == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(config_id=WORKLOADS[args.workload][0], max_procs=args.cpu_procs)

    rank, world, device, shared, backend = dist_setup(args)
    runner = run_plate if args.workload in PLATE_WORKLOADS else run_resident
    line = runner(args, rank, world, device, shared, backend, cpu)
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
