"""The RCCL code paths on real hardware: they switch on when the box shows two or more GPUs.

The build box and the round-end test box have one GPU, where these are skipped (the world-size-2
``gloo`` tests in ``test_pipeline.py`` / ``test_slab.py`` cover the same logic on the CPU); on an
8-GPU node they run:

* units sharded over ranks + ``gather_to_rank0`` over ``nccl`` == the single-rank results;
* the sharded ``reconstruct`` CLI over N ranks writes the store a single rank writes;
* one volume split into row slabs over 2 and 4 ranks, halos exchanged GPU to GPU, == the unsplit run;
* ``bench.py --gpus 2`` prints its line with RCCL carrying the barrier.

Each case is its own ``torch.distributed.run`` launch (fresh processes, one rank per GPU,
127.0.0.1 rendezvous).
"""

import json
import os
import socket
import subprocess
import sys

from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _n_gpus() -> int:
    import torch

    return torch.cuda.device_count()      # (does not initialise the GPU)


def _need(n):
    if _n_gpus() < n:
        pytest.skip(f"needs {n} GPUs, this box shows {_n_gpus()}")


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(nproc, *argv, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), *argv]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, f"{' '.join(cmd)}\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}"
    return r.stdout


def _json_line(out, key):
    for line in reversed(out.splitlines()):
        if line.startswith("{") and key in line:
            return json.loads(line)
    raise AssertionError(f"no JSON line with {key!r} in:\n{out[-2000:]}")


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_units_and_rccl_gather_equal_the_single_rank_results(world):
    _need(world)
    rec = _json_line(_launch(world, "tools/sharded_gather_check.py", "--units", "7"), "sharded_gather")
    assert rec["backend"] == "nccl" and rec["world_size"] == world and rec["equal_to_single_rank"]
    assert sum(rec["units_per_rank"]) == 7 and max(rec["units_per_rank"]) - min(rec["units_per_rank"]) <= 1


@pytest.mark.parametrize("world", [2, 4])
def test_row_slab_split_over_real_ranks_is_bit_equal_to_the_unsplit_run(world):
    """``SlabRichardsonLucy`` with device-to-device halo sends (``slab.exchange_halos`` on RCCL)."""
    _need(world)
    out = _launch(world, "tools/slab_check.py", "--backend", "nccl", "--shape", "40,256,300", "--iters", "8")
    assert f"over {world} ranks (nccl): equal to the unsplit run: True" in out


def test_sharded_cli_over_two_gpus_writes_the_single_rank_store(tmp_path):
    _need(2)
    import yaml

    from shrimpy_amd.io.omezarr import open_ome_zarr

    rng = np.random.default_rng(1)
    with open_ome_zarr(tmp_path / "raw.zarr", layout="hcs", mode="w", channel_names=["LS"],
                       prefer_iohub=False) as plate:
        for i in range(5):
            arr = plate.create_position("A", str(i + 1), "0").create_zeros(
                "0", shape=(2, 1, 96, 24, 70), dtype="uint16", scale=(1, 1, 0.15, 0.1133, 0.1133))
            for t in range(2):
                arr.write_volume(t, 0, rng.integers(90, 900, (96, 24, 70)).astype(np.uint16))
    (tmp_path / "recon.yml").write_text(yaml.safe_dump(dict(
        deskew=dict(pixel_size_um=0.1133, ls_angle_deg=30.0, scan_step_um=0.15, keep_overhang=False,
                    average_n_slices=3),
        deconvolution=dict(iterations=5, gaussian_shape_zyx=[5, 5, 5], gaussian_sigma_zyx=[1.2, 1.0, 1.0]))))
    args = ["reconstruct", "-i", str(tmp_path / "raw.zarr"), "-c", str(tmp_path / "recon.yml")]
    _launch(1, "-m", "shrimpy_amd", *args, "-o", str(tmp_path / "one.zarr"))
    out = _launch(2, "-m", "shrimpy_amd", *args, "-o", str(tmp_path / "two.zarr"))
    assert "'world_size': 2" in out
    with open_ome_zarr(tmp_path / "one.zarr", prefer_iohub=False) as a, \
            open_ome_zarr(tmp_path / "two.zarr", prefer_iohub=False) as b:
        for (ka, pa), (kb, pb) in zip(a.positions(), b.positions()):
            assert ka == kb
            np.testing.assert_array_equal(pa["0"][:], pb["0"][:])
    # a second two-rank launch onto the existing store stops on EVERY rank with the same message
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "-m", "shrimpy_amd",
                        *args, "-o", str(tmp_path / "two.zarr")], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "--resume" in (r.stdout + r.stderr)


def test_bench_with_two_gpus_runs_over_rccl():
    _need(2)
    out = _launch(2, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "small")
    line = _json_line(out, '"metric"')
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert line["config"]["collective_backend"] == "nccl"
    assert "REHEARSAL" not in line["config"]["parallelism"]
    assert line["value"] > 0 and "cpu_baseline" not in line
