"""BASELINE configs 4 and 5 through ``cli.run_store`` at their real per-unit sizes, on one GPU.

* config 4: a plate of raw ``(2048, 256, 2048)`` uint16 positions, deskew + 20-iteration RL, the
  pinned staging path -- each position checked against the oracle the way
  ``tests/test_full_size_gpu.py`` does: the deskew bit for bit on raw-X slabs, the stored RL result on a
  crop with the full domain of dependence.
* config 5: raw ``(2048, 200, 2048)`` with T x P streamed units, deskew -> register -> deconvolve;
  every unit is read and written exactly once, in unit order, and each stored volume is the
  reconstruction of ITS (position, timepoint) -- checked against the oracle chain on a crop.

Seeds follow SURVEY.md section 8(d): ``1000 * config + 7 * position + timepoint``.  The plates live under
pytest's tmp_path (about 16 GB and 19 GB); a box without that much scratch space skips with the
reason stated.
"""

import shutil

import numpy as np
import pytest

from oracle import cpu_ref as o

pytestmark = pytest.mark.gpu

DESKEW = dict(ls_angle_deg=30.0, px_to_scan_ratio=0.755, keep_overhang=False, average_n_slices=3)
PSF_SHAPE, PSF_SIGMA = (9, 7, 7), (2.0, 1.2, 1.2)
ITERS = 20
MARGIN, CORE = 120, 32          # 20 iterations x 2 stencils x radius 3 in the plane; all of z is kept


def _need_scratch(tmp_path, gib):
    free = shutil.disk_usage(tmp_path).free / 2**30
    if free < gib:
        pytest.skip(f"{tmp_path} has {free:.0f} GiB free, this plate needs {gib} GiB")


def _settings(register=None):
    from shrimpy_amd.settings import DeconvolveSettings, DeskewSettings, ReconstructSettings, RegisterSettings

    return ReconstructSettings(
        deskew=DeskewSettings(pixel_size_um=0.1133, scan_step_um=0.15, ls_angle_deg=30.0, keep_overhang=False,
                              average_n_slices=3),
        registration=None if register is None else RegisterSettings(affine_transform_zyx=register.tolist()),
        deconvolution=DeconvolveSettings(iterations=ITERS))


def _write_plate(path, raw_shape, keys, nt, config, device):
    """uint16 bead scenes, one per (position, t); acquisition chunking (1, 1, 32, ny, nx)."""
    import torch

    import bench
    from shrimpy_amd.io.omezarr import open_ome_zarr

    with open_ome_zarr(path, layout="hcs", mode="w", channel_names=["LS"], prefer_iohub=False) as plate:
        for p, key in enumerate(keys):
            arr = plate.create_position(*key.split("/")).create_zeros(
                "0", shape=(nt, 1) + tuple(raw_shape), dtype="uint16", scale=(1, 1, 0.15, 0.1133, 0.1133))
            for t in range(nt):
                raw = bench.synthetic_raw(raw_shape, seed=1000 * config + 7 * p + t, device=device)
                assert float(raw.max()) < 65536
                arr.write_volume(t, 0, raw.to(torch.uint16).cpu().numpy())
                del raw
    torch.cuda.empty_cache()


def _log_volume_io(monkeypatch):
    """Record every whole-volume read / write the run makes, in order."""
    from shrimpy_amd.io.omezarr import ZarrArray

    reads, writes = [], []
    real_read, real_write = ZarrArray.read_volume, ZarrArray.write_volume

    def key_of(arr):
        return "/".join(arr.path.parts[-4:-1])

    def read(self, *lead, out=None):
        reads.append((key_of(self),) + tuple(lead))
        return real_read(self, *lead, out=out)

    def write(self, *args):
        writes.append((key_of(self),) + tuple(args[:-1]))
        return real_write(self, *args)

    monkeypatch.setattr(ZarrArray, "read_volume", read)
    monkeypatch.setattr(ZarrArray, "write_volume", write)
    return reads, writes


def _check_deskew_on_slabs(raw16, deskewed):
    """The deskew of this raw shape, bit for bit, on raw-X column slabs (first, middle, last)."""
    X = raw16.shape[2]
    for a, b in ((0, 2), (X // 2 - 1, X // 2 + 2), (X - 3, X)):
        slab = raw16[:, :, a:b].cpu().numpy().astype(np.float32)
        want = o.deskew(slab, DESKEW["ls_angle_deg"], DESKEW["px_to_scan_ratio"], DESKEW["keep_overhang"],
                        DESKEW["average_n_slices"])
        np.testing.assert_array_equal(deskewed[:, X - b:X - a, :].cpu().numpy(), want)


def _rl_crop_matches(y_crop, stored_core, factors):
    want = o.richardson_lucy_separable(y_crop, factors, iterations=ITERS)[
        :, MARGIN:MARGIN + CORE, MARGIN:MARGIN + CORE].astype(np.float64)
    got = stored_core.astype(np.float64)
    tol = 2e-4 * np.abs(want) + 1e-4 * np.abs(want).max()        # the RL bar of tests/test_gpu_parity.py
    assert np.all(np.abs(got - want) <= tol), float(np.max(np.abs(got - want) / tol))


def test_config4_plate_deskew_rl_through_the_staged_store_path(tmp_path, device, monkeypatch):
    import torch

    import shrimpy_amd.staging as staging
    from shrimpy_amd import cli
    from shrimpy_amd.deskew import fast_deskew_zyx
    from shrimpy_amd.io.omezarr import open_ome_zarr

    _need_scratch(tmp_path, 20)
    raw_shape = (2048, 256, 2048)
    keys = ["A/1/0", "A/2/0", "B/1/0", "B/2/0"]
    _write_plate(tmp_path / "plate.zarr", raw_shape, keys, 1, 4, device)
    stagers = []
    real_init = staging.VolumeStager.__init__

    def spy_init(self, *a, **k):
        real_init(self, *a, **k)
        stagers.append(self)

    monkeypatch.setattr(staging.VolumeStager, "__init__", spy_init)
    reads, writes = _log_volume_io(monkeypatch)
    res = cli.run_store(tmp_path / "plate.zarr", tmp_path / "recon.zarr", _settings())
    assert res["units"] == res["units_total"] == 4 and res["output_shape"] == (86, 2048, 2491)
    assert len(stagers) == 1 and stagers[0].raw_shape == raw_shape       # the pinned staging path ran
    assert reads == [(k, 0, 0) for k in keys] and writes == reads
    factors = o.gaussian_psf(PSF_SHAPE, PSF_SIGMA)[1]
    with open_ome_zarr(tmp_path / "plate.zarr", prefer_iohub=False) as src, \
            open_ome_zarr(tmp_path / "recon.zarr", prefer_iohub=False) as dst:
        for p, key in enumerate(keys):
            raw16 = torch.as_tensor(src[key]["0"].read_volume(0, 0), device=device)
            deskewed = fast_deskew_zyx(raw_data=raw16, **DESKEW)
            assert tuple(deskewed.shape) == (86, 2048, 2491)
            _check_deskew_on_slabs(raw16, deskewed)
            out = dst[key]["0"].read_volume(0, 0)
            assert out.dtype == np.float32 and np.isfinite(out).all() and out.min() >= 0
            # a different crop per position: interior ones and the volume's own corner
            y0, x0 = [(900, 1300), (MARGIN, MARGIN), (1500, 300), (0, 0)][p]
            if (y0, x0) == (0, 0):   # real borders on two sides
                crop = (slice(None), slice(0, CORE + MARGIN), slice(0, CORE + MARGIN))
                want = o.richardson_lucy_separable(deskewed[crop].contiguous().cpu().numpy(), factors,
                                                   iterations=ITERS)[:, :CORE, :CORE].astype(np.float64)
                got = out[:, :CORE, :CORE].astype(np.float64)
                assert np.all(np.abs(got - want) <= 2e-4 * np.abs(want) + 1e-4 * np.abs(want).max())
            else:
                crop = (slice(None), slice(y0 - MARGIN, y0 + CORE + MARGIN), slice(x0 - MARGIN, x0 + CORE + MARGIN))
                _rl_crop_matches(deskewed[crop].contiguous().cpu().numpy(),
                                 out[:, y0:y0 + CORE, x0:x0 + CORE], factors)
            del raw16, deskewed
    torch.cuda.empty_cache()
    shutil.rmtree(tmp_path / "plate.zarr", ignore_errors=True)     # 16 GB: do not leave them for the session
    shutil.rmtree(tmp_path / "recon.zarr", ignore_errors=True)


def _config5_matrix():
    """The registration of SURVEY 8(d) config 3 (rotation 2 deg about Z, scale (1, .98, 1.02),
    translation (3.5, -12.25, 20.75)), applied to every streamed unit."""
    th = np.deg2rad(2.0)
    rot = np.array([[1, 0, 0], [0, np.cos(th), -np.sin(th)], [0, np.sin(th), np.cos(th)]])
    m = np.eye(4)
    m[:3, :3] = rot @ np.diag([1.0, 0.98, 1.02])
    m[:3, 3] = [3.5, -12.25, 20.75]
    return m


def _oracle_register_block(deskewed, m, origin, size):
    """The registered volume on an output block, from the source box it reaches (as in
    ``tests/test_full_size_gpu.py``: block-local fp64 coordinates, so at most one ulp off)."""
    shape = np.array(deskewed.shape)
    origin, size = np.array(origin), np.array(size)
    corners = np.array([[origin[i] + (size[i] - 1) * ((c >> i) & 1) for i in range(3)] for c in range(8)])
    src = corners @ m[:3, :3].T + m[:3, 3]
    lo = np.maximum(np.floor(src.min(0)).astype(int) - 2, 0)
    hi = np.minimum(np.ceil(src.max(0)).astype(int) + 3, shape)
    crop = deskewed[tuple(slice(a, b) for a, b in zip(lo, hi))].contiguous().cpu().numpy()
    offset = m[:3, :3] @ origin + m[:3, 3] - lo
    return o.affine_apply(crop, m[:3, :3], offset, tuple(size))


def test_config5_streamed_units_deskew_register_deconvolve(tmp_path, device, monkeypatch):
    import torch

    import shrimpy_amd.staging as staging
    from shrimpy_amd import cli
    from shrimpy_amd.deskew import fast_deskew_zyx
    from shrimpy_amd.io.omezarr import open_ome_zarr
    from shrimpy_amd.pipeline import enumerate_units

    _need_scratch(tmp_path, 24)
    raw_shape = (2048, 200, 2048)
    keys, nt = ["0/1/000", "0/2/000"], 3
    _write_plate(tmp_path / "lapse.zarr", raw_shape, keys, nt, 5, device)
    m = _config5_matrix()
    staged = []
    real_stage_in = staging.VolumeStager.stage_in

    def spy_stage_in(self, slot, data=None):
        staged.append(slot)
        return real_stage_in(self, slot, data)

    monkeypatch.setattr(staging.VolumeStager, "stage_in", spy_stage_in)
    reads, writes = _log_volume_io(monkeypatch)
    res = cli.run_store(tmp_path / "lapse.zarr", tmp_path / "recon.zarr", _settings(register=m), zarr_version="0.5")
    units = [(u.position, u.t, u.c) for u in enumerate_units(keys, nt, range(1))]
    assert res["units"] == res["units_total"] == len(units) == 6 and res["output_shape"] == (67, 2048, 2540)
    # streamed through the stager: every unit once, in unit order, alternating slots; none dropped
    assert reads == units and writes == units
    assert staged == [i % 2 for i in range(6)]
    factors = o.gaussian_psf(PSF_SHAPE, PSF_SIGMA)[1]
    y0, x0 = 1000, 1200
    origin = (0, y0 - MARGIN, x0 - MARGIN)
    size = (67, CORE + 2 * MARGIN, CORE + 2 * MARGIN)
    cores = {}
    with open_ome_zarr(tmp_path / "lapse.zarr", prefer_iohub=False) as src, \
            open_ome_zarr(tmp_path / "recon.zarr", prefer_iohub=False) as dst:
        for key, t, _ in units:
            raw16 = torch.as_tensor(src[key]["0"].read_volume(t, 0), device=device)
            deskewed = fast_deskew_zyx(raw_data=raw16, **DESKEW)
            if t == 0:
                _check_deskew_on_slabs(raw16, deskewed)
            registered = _oracle_register_block(deskewed, m, origin, size)
            core = dst[key]["0"].read_volume(t, 0)[:, y0:y0 + CORE, x0:x0 + CORE]
            _rl_crop_matches(registered, core, factors)
            cores[key, t] = core
            del raw16, deskewed
    # every unit holds its own scene (different seeds): no two stored cores coincide
    ks = list(cores)
    for i in range(len(ks)):
        for j in range(i + 1, len(ks)):
            assert not np.array_equal(cores[ks[i]], cores[ks[j]])
    torch.cuda.empty_cache()
    shutil.rmtree(tmp_path / "lapse.zarr", ignore_errors=True)
    shutil.rmtree(tmp_path / "recon.zarr", ignore_errors=True)
