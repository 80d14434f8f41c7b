"""Sharding of independent (position, timepoint, channel) units over ranks.

The N > 1 path is exercised with real ``torch.distributed`` processes (gloo, world_size 2, CPU):
the partition, the barrier + max-over-ranks timing and the optional gather to rank 0.  The
compute callable is a stand-in here (no GPU in the build container); the GPU test at the bottom
runs the real ``VolumeReconstructor`` on one device.
"""

import os
import socket

import numpy as np
import pytest

from shrimpy_amd.pipeline import Unit, enumerate_units, shard_units


def test_enumerate_units_order():
    units = enumerate_units(["A/1/0", "A/2/0"], n_t=2, channels=(0, 1))
    assert len(units) == 8
    assert units[0] == Unit("A/1/0", 0, 0) and units[1] == Unit("A/1/0", 0, 1)
    assert units[4] == Unit("A/2/0", 0, 0)


@pytest.mark.parametrize("n,world", [(96, 8), (96, 1), (5, 8), (0, 4), (19200, 8), (7, 2)])
def test_shard_units_is_a_balanced_partition(n, world):
    units = list(range(n))
    shards = [shard_units(units, r, world) for r in range(world)]
    assert sorted(u for s in shards for u in s) == units          # every unit exactly once
    sizes = [len(s) for s in shards]
    assert max(sizes) - min(sizes) <= 1                           # 96 positions / 8 GPUs -> 12 each
    assert shards[0][:2] == units[0:2 * world:world]


def test_shard_units_rejects_bad_rank():
    with pytest.raises(ValueError):
        shard_units([1, 2], 2, 2)
    with pytest.raises(ValueError):
        shard_units([1, 2], 0, 0)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmp):
    import torch
    import torch.distributed as dist

    from shrimpy_amd.pipeline import gather_to_rank0, run_sharded

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        units = enumerate_units([f"A/{i}/0" for i in range(7)])
        results = {}

        def load(u):
            return torch.full((2, 3, 4), float(int(u.position.split("/")[1])))

        def process(v):
            return v * 2 + 1

        def store(u, v):
            results[u] = v

        rep = run_sharded(units, load, process, store)
        assert rep.world_size == world and rep.rank == rank
        assert rep.units == shard_units(units, rank, world)
        assert rep.max_seconds >= rep.seconds > 0
        gathered = gather_to_rank0([results[u] for u in rep.units], len(units))
        if rank == 0:
            vals = [float(t[0, 0, 0]) for t in gathered]
            np.save(os.path.join(tmp, "gathered.npy"), np.array(vals))
        else:
            assert gathered is None
        with open(os.path.join(tmp, f"ok{rank}"), "w") as f:
            f.write(",".join(u.position for u in rep.units))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_run_sharded_world_size_2_gloo(tmp_path, world):
    """Seven units over 2 and over 3 ranks (uneven shares: 4 + 3, 3 + 2 + 2), real processes over gloo."""
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    owned = [(tmp_path / f"ok{r}").read_text().split(",") for r in range(world)]
    for r in range(world):
        assert owned[r] == [f"A/{i}/0" for i in range(r, 7, world)]
    # rank 0 holds every unit's result, in unit order
    np.testing.assert_array_equal(np.load(tmp_path / "gathered.npy"), [2 * i + 1 for i in range(7)])


def _gather_worker(rank, world, port, tmp):
    import torch
    import torch.distributed as dist

    from shrimpy_amd.pipeline import gather_to_rank0

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_total = 8                                            # 3 + 3 + 2 units: unequal shares
        mine = list(range(rank, n_total, world))
        # every unit its own shape and value (the gather may not assume rank 0's first shape)
        local = [torch.full((2 + u, 3, 1 + u % 3), float(u), dtype=torch.float32 if u % 2 else torch.float64) for u in mine]
        for keep in (False, True):
            got = gather_to_rank0(local, n_total, keep_on_host=keep)
            if rank == 0:
                assert [tuple(t.shape) for t in got] == [(2 + u, 3, 1 + u % 3) for u in range(n_total)]
                assert all(float(t.flatten()[0]) == u and bool((t == u).all()) for u, t in enumerate(got))
                assert [t.dtype for t in got] == [torch.float32 if u % 2 else torch.float64 for u in range(n_total)]
            else:
                assert got is None
        # a rank that holds the wrong number of results is reported on every rank, nobody hangs
        try:
            gather_to_rank0(local[:-1] if rank == 1 else local, n_total)
            raise AssertionError("expected a RuntimeError")
        except RuntimeError as exc:
            assert "round-robin" in str(exc)
        with open(os.path.join(tmp, f"gok{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


def test_gather_to_rank0_unequal_counts_and_shapes_world_3_gloo(tmp_path):
    """Round-4 verdict: the gather posted one blocking receive at a time and assumed every unit had rank 0's first shape.
    Three ranks, 3 + 3 + 2 units, every unit a different shape and dtype; every receive of a round is posted at once."""
    import torch.multiprocessing as mp

    mp.spawn(_gather_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    assert all((tmp_path / f"gok{r}").read_text() == "ok" for r in range(3))


def test_run_sharded_skips_failed_units_and_reports_them():
    """``on_error="skip"`` (the reference keeps acquiring after a failed stack, shrimpy/dynatrack/worker.py:262-271): a unit
    whose load, process or store raises is left out and listed; every other unit is stored."""
    from shrimpy_amd.pipeline import run_sharded

    out = {}

    def load(u):
        if u == 2:
            raise OSError("chunk 2 is damaged")
        return u

    def process(v):
        if v == 4:
            raise ValueError("bad volume")
        return v * v

    def store(u, v):
        if u == 6:
            raise OSError("disk full")
        out[u] = v

    rep = run_sharded(list(range(8)), load, process, store, on_error="skip")
    assert out == {u: u * u for u in (0, 1, 3, 5, 7)}
    assert [(u, stage) for u, stage, _ in rep.failures] == [(2, "load"), (4, "process"), (6, "store")]
    assert "damaged" in rep.failures[0][2] and "OSError" in rep.failures[0][2]
    with pytest.raises(OSError):
        run_sharded(list(range(8)), load, process, store)          # the default still raises
    with pytest.raises(ValueError):
        run_sharded([1], load, process, store, on_error="ignore")


def test_run_sharded_single_process():
    from shrimpy_amd.pipeline import gather_to_rank0, run_sharded

    out = {}
    rep = run_sharded(list(range(5)), lambda u: u, lambda v: v * v, lambda u, v: out.__setitem__(u, v))
    assert rep.world_size == 1 and out == {i: i * i for i in range(5)}
    assert gather_to_rank0([1, 2, 3], 3) == [1, 2, 3]


@pytest.mark.gpu
def test_volume_reconstructor_matches_oracle(device):
    """The whole per-volume path on one GPU vs the oracle, small size."""
    import torch

    from oracle import cpu_ref as o
    from shrimpy_amd.pipeline import VolumeReconstructor
    from shrimpy_amd.settings import DeconvolveSettings, DeskewSettings, ReconstructSettings, RegisterSettings

    psf, factors = o.gaussian_psf((9, 7, 7), (2.0, 1.2, 1.2))
    raw = o.bead_scene((96, 24, 80), seed=77, psf=psf, density=1e-3)
    m = np.eye(4)
    m[:3, 3] = [0.25, -1.5, 2.0]
    settings = ReconstructSettings(
        deskew=DeskewSettings(pixel_size_um=0.1133, ls_angle_deg=30, scan_step_um=0.15),
        registration=RegisterSettings(affine_transform_zyx=m.tolist()),
        deconvolution=DeconvolveSettings(iterations=10),
    )
    rec = VolumeReconstructor(raw.shape, settings, device)
    out = rec(raw).cpu().numpy().astype(np.float64)
    d = o.deskew(raw, 30.0, 0.755, False, 3)
    assert rec.output_shape == d.shape
    a = o.affine_apply_4x4(d, m, d.shape)
    ref = o.richardson_lucy(a, psf, 10).astype(np.float64)
    tol = 1e-4 * np.abs(ref) + 5e-5 * np.abs(ref).max()
    assert np.all(np.abs(out - ref) <= tol)
    assert torch.cuda.is_available()


@pytest.mark.gpu
def test_volume_reconstructor_with_flatfield_matches_oracle(device):
    """flatfield: true -> the median kernel, its division fused into the deskew, then RL."""
    from oracle import cpu_ref as o
    from shrimpy_amd.pipeline import VolumeReconstructor
    from shrimpy_amd.settings import DeconvolveSettings, DeskewSettings, ReconstructSettings

    psf, _ = o.gaussian_psf((9, 7, 7), (2.0, 1.2, 1.2))
    rng = np.random.default_rng(9)
    raw = (o.bead_scene((96, 24, 80), seed=78, psf=psf, density=1e-3)
           * (0.7 + 0.6 * rng.random((1, 24, 80)))).astype(np.float32)       # uneven illumination
    settings = ReconstructSettings(
        flatfield=True,
        deskew=DeskewSettings(pixel_size_um=0.1133, ls_angle_deg=30, scan_step_um=0.15),
        deconvolution=DeconvolveSettings(iterations=5),
    )
    out = VolumeReconstructor(raw.shape, settings, device)(raw).cpu().numpy().astype(np.float64)
    ref = o.richardson_lucy(o.deskew(o.flat_field_bf(raw), 30.0, 0.755, False, 3), psf, 5).astype(np.float64)
    assert np.all(np.abs(out - ref) <= 1e-4 * np.abs(ref) + 5e-5 * np.abs(ref).max())
    only = VolumeReconstructor(raw.shape, ReconstructSettings(flatfield=True), device)(raw).cpu().numpy()
    np.testing.assert_allclose(only, o.flat_field_bf(raw), rtol=2e-6)
    # uint16 stacks: uploaded unconverted, same result as their float32 copy
    import torch

    counts = rng.integers(0, 4000, raw.shape).astype(np.uint16)
    s16 = ReconstructSettings(deskew=DeskewSettings(pixel_size_um=0.1133, ls_angle_deg=30, scan_step_um=0.15),
                              deconvolution=DeconvolveSettings(iterations=3))
    rec = VolumeReconstructor(raw.shape, s16, device)
    assert torch.equal(rec(counts).clone(), rec(counts.astype(np.float32)))


def test_run_sharded_overlaps_io_with_compute_and_keeps_order():
    """Loads run one unit ahead and stores one unit behind, on background threads; results are
    the same as the serial loop and errors in a store surface."""
    import threading
    import time as _t

    from shrimpy_amd.pipeline import run_sharded

    events, out = [], {}
    main = threading.get_ident()

    def load(u):
        events.append(("load", u, threading.get_ident() != main))
        _t.sleep(0.02)
        return u

    def process(v):
        events.append(("process", v, threading.get_ident() == main))
        _t.sleep(0.02)
        return v * 10

    def store(u, v):
        events.append(("store", u, threading.get_ident() != main))
        out[u] = v

    rep = run_sharded(list(range(6)), load, process, store)
    assert out == {i: i * 10 for i in range(6)} and rep.units == list(range(6))
    assert all(flag for _, _, flag in events)            # loads/stores off-thread, process on-thread
    order = [(k, u) for k, u, _ in events]
    assert order.index(("load", 1)) < order.index(("store", 0))  # unit 1 was being read before 0 was written
    serial = {}
    run_sharded(list(range(6)), lambda u: u, lambda v: v * 10, lambda u, v: serial.__setitem__(u, v),
                overlap_io=False)
    assert serial == out

    def bad_store(u, v):
        raise OSError("disk full")

    with pytest.raises(OSError, match="disk full"):
        run_sharded(list(range(3)), lambda u: u, lambda v: v, bad_store)


@pytest.mark.gpu
@pytest.mark.parametrize("in_place", [True, False])
def test_run_sharded_staged_through_pinned_slots_matches_direct_calls(device, in_place):
    """Upload / kernels / download on three streams with two slots: every unit's result equals the
    synchronous call's, in order, whether the loader fills the pinned slot or returns its own array."""
    import torch

    from shrimpy_amd.pipeline import VolumeReconstructor, run_sharded
    from shrimpy_amd.settings import DeconvolveSettings, DeskewSettings, ReconstructSettings
    from shrimpy_amd.staging import VolumeStager

    rng = np.random.default_rng(21)
    raw_shape = (96, 24, 70)
    vols = [rng.integers(90, 900, raw_shape).astype(np.uint16) for _ in range(7)]
    settings = ReconstructSettings(
        deskew=DeskewSettings(pixel_size_um=0.1133, ls_angle_deg=30.0, px_to_scan_ratio=0.755,
                              keep_overhang=False, average_n_slices=3),
        deconvolution=DeconvolveSettings(iterations=4, gaussian_shape_zyx=(5, 5, 5), gaussian_sigma_zyx=(1.2, 1.0, 1.0)),
        flatfield=True)
    rec = VolumeReconstructor(raw_shape, settings, device)
    want = [rec(v).cpu().numpy() for v in vols]
    stager = VolumeStager(raw_shape, np.uint16, rec.output_shape, device)
    got = {}

    if in_place:
        def load(u, out=None):
            out[...] = vols[u]
            return out
    else:
        def load(u):
            return vols[u]

    rep = run_sharded(list(range(len(vols))), load, rec, lambda u, v: got.__setitem__(u, v.copy()),
                      synchronize=torch.cuda.synchronize, stager=stager)
    assert rep.units == list(range(len(vols)))
    for u, w in enumerate(want):
        np.testing.assert_array_equal(got[u], w)


def test_stager_refuses_cpu_devices():
    from shrimpy_amd._lib import LsrError
    from shrimpy_amd.staging import VolumeStager

    with pytest.raises(LsrError, match="needs a HIP device"):
        VolumeStager((4, 4, 4), np.uint16, (4, 4, 4), "cpu")


@pytest.mark.gpu
def test_staging_slots_are_driver_owned_pinned_memory_of_the_exact_size(device):
    """``lsr_pinned_alloc`` (hipHostMalloc) behind a torch tensor: page-locked as torch sees it, exactly the
    slot's size, asynchronous copies both ways, and valid for as long as any view of it lives."""
    import gc

    import torch

    from shrimpy_amd.staging import VolumeStager, _pinned_tensor

    t = _pinned_tensor((3, 5, 7), torch.uint16)
    assert t.is_pinned() and t.is_contiguous() and t.numel() * t.element_size() == 3 * 5 * 7 * 2
    t.numpy()[...] = np.arange(105, dtype=np.uint16).reshape(3, 5, 7)
    d = torch.empty((3, 5, 7), dtype=torch.uint16, device=device)
    d.copy_(t, non_blocking=True)
    back = _pinned_tensor((3, 5, 7), torch.uint16)
    back.copy_(d, non_blocking=True)
    torch.cuda.synchronize()
    assert torch.equal(back, t)
    view = back.numpy()
    del back
    gc.collect()
    assert view.sum() == np.arange(105).sum()          # the view keeps the allocation alive

    st = VolumeStager((16, 8, 12), np.uint16, (4, 12, 20), device)
    assert all(s.is_pinned() for s in st._host_in + st._host_out)
    held = st.host_in(0)
    held[...] = 7
    st.close()
    assert int(held.sum()) == 7 * 16 * 8 * 12          # closed, but a caller's view stays readable
