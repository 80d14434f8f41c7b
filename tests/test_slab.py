"""Single-volume row-slab split (SURVEY 8 e / f-4): slab geometry, the halo exchange over a
world-size-2 gloo group on the CPU, and -- on the GPU -- equality of the slab-wise RL (all ranks
emulated in one process) with the single-GPU result."""

from __future__ import annotations

import os
import sys

from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def test_slab_ranges_cover_the_rows_once_with_clipped_halos():
    from shrimpy_amd.slab import slab_ranges

    for n, world, halo in ((2048, 8, 6), (100, 3, 6), (37, 2, 14), (10, 1, 6)):
        slabs = slab_ranges(n, world, halo)
        assert [s.own0 for s in slabs] == [0] + [s.own1 for s in slabs[:-1]] and slabs[-1].own1 == n
        assert max(s.own1 - s.own0 for s in slabs) - min(s.own1 - s.own0 for s in slabs) <= 1
        for s in slabs:
            assert s.ext0 == max(0, s.own0 - halo) and s.ext1 == min(n, s.own1 + halo)
            assert s.rows == s.lo + (s.own1 - s.own0) + s.hi
    with pytest.raises(ValueError, match="thinner than the halo"):
        slab_ranges(20, 4, 6)
    with pytest.raises(ValueError):
        slab_ranges(3, 4, 0)


def _exchange_worker(rank, world, port, n_rows, halo, tmp):
    import torch
    import torch.distributed as dist

    from shrimpy_amd.slab import exchange_halos, slab_ranges

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        slab = slab_ranges(n_rows, world, halo)[rank]
        # every row carries its GLOBAL index in the owned part, -1 in the halos
        view = torch.full((3, slab.rows, 5), -1.0)
        view[:, slab.lo:slab.rows - slab.hi, :] = torch.arange(slab.own0, slab.own1, dtype=torch.float32)[None, :, None]
        exchange_halos(view, slab)
        want = torch.arange(slab.ext0, slab.ext1, dtype=torch.float32)[None, :, None].expand(3, -1, 5)
        np.save(os.path.join(tmp, f"ok{rank}.npy"), np.array([bool(torch.equal(view, want))]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_exchange_halos_world_size_n_gloo(tmp_path, world):
    import torch.multiprocessing as mp

    port = 29600 + world + os.getpid() % 200
    mp.spawn(_exchange_worker, args=(world, port, 41, 6, str(tmp_path)), nprocs=world, join=True)
    assert all(bool(np.load(tmp_path / f"ok{r}.npy")[0]) for r in range(world))


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_slab_rl_equals_single_gpu_rl_bit_exact(world):
    """Every voxel sees the same inputs and the same arithmetic as in the unsplit run."""
    import torch

    from oracle import cpu_ref as o
    from shrimpy_amd.deconvolve import RichardsonLucyPlan
    from shrimpy_amd.slab import SlabRichardsonLucy, run_slabs_in_process

    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    dev = torch.device("cuda:0")
    psf, factors = o.gaussian_psf((9, 7, 7), (2.0, 1.2, 1.2))
    y = o.bead_scene((20, 70, 150), seed=11, psf=psf, density=2e-3)
    yt = torch.as_tensor(y, device=dev)
    whole = RichardsonLucyPlan(y.shape, None, dev, psf_factors=factors)(yt, iterations=6)
    slabs = [SlabRichardsonLucy(y.shape, factors, dev, r, world) for r in range(world)]
    for s in slabs:
        s.y_pad.view.copy_(yt[:, s.slab.ext0:s.slab.ext1, :])
    parts = run_slabs_in_process(slabs, iterations=6)
    assert torch.equal(torch.cat(parts, dim=1), whole)


@pytest.mark.gpu
def test_deskew_slab_writes_the_same_rows_as_the_whole_deskew():
    import torch

    from oracle import cpu_ref as o
    from shrimpy_amd.deskew import fast_deskew_zyx
    from shrimpy_amd.slab import SlabRichardsonLucy, deskew_slab

    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    dev = torch.device("cuda:0")
    _, factors = o.gaussian_psf((9, 7, 7), (2.0, 1.2, 1.2))
    raw = torch.as_tensor(np.random.default_rng(4).integers(80, 600, (140, 30, 100)).astype(np.float32), device=dev)
    kw = dict(ls_angle_deg=30.0, px_to_scan_ratio=0.755, keep_overhang=False, average_n_slices=3)
    whole = fast_deskew_zyx(raw_data=raw, **kw)
    x = raw.shape[2]
    for rank in range(2):
        s = SlabRichardsonLucy(tuple(whole.shape), factors, dev, rank, 2)
        deskew_slab(raw[:, :, x - s.slab.ext1:x - s.slab.ext0].contiguous(), s, **kw)
        assert torch.equal(s.y_pad.view, whole[:, s.slab.ext0:s.slab.ext1, :])


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_distributed_slab_run_with_ranks_sharing_the_card_equals_the_unsplit_run(world):
    """The REAL distributed path -- ``torch.distributed.run`` ranks, ``SlabRichardsonLucy.run`` with
    ``exchange_halos`` as batched point-to-point sends after every iteration -- on the one GPU of the test box: the
    ranks share the card and meet over gloo (halo rows staged through the host; RCCL refuses duplicate devices).
    Same kernels, same protocol as the RCCL run of ``tests/test_multi_gpu.py``, which needs one GPU per rank."""
    import subprocess
    import sys

    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    from tests.test_multi_gpu import ROOT, _free_port

    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "tools/slab_check.py",
           "--backend", "gloo", "--shape", "24,96,200", "--iters", "6"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, f"{r.stdout[-2000:]}\n{r.stderr[-3000:]}"
    assert f"over {world} ranks (gloo): equal to the unsplit run: True" in r.stdout


@pytest.mark.gpu
def test_sharded_units_and_gather_with_ranks_sharing_the_card_equal_the_single_rank_results():
    """``run_sharded`` + ``gather_to_rank0`` under ``torch.distributed.run`` with three ranks on the one GPU (gloo,
    results staged through the host): rank 0 recomputes every unit and compares bit for bit."""
    import json
    import subprocess
    import sys

    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    from tests.test_multi_gpu import ROOT, _free_port

    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "tools/sharded_gather_check.py",
           "--backend", "gloo", "--units", "7"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, f"{r.stdout[-2000:]}\n{r.stderr[-3000:]}"
    rec = next(json.loads(ln) for ln in reversed(r.stdout.splitlines()) if ln.startswith("{") and "sharded_gather" in ln)
    assert rec["backend"] == "gloo" and rec["world_size"] == 3 and rec["equal_to_single_rank"]
    assert rec["units_per_rank"] == [3, 2, 2]
