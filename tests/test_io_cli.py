"""OME-Zarr in/out and the deskew / register / deconvolve CLI.

Fixtures follow the reference's ``shrimpy/tests/test_replay_camera.py``: synthetic stores in
``tmp_path`` whose pixel values encode ``(p, t, c, z)`` (``:33-47``), a FOV and a 2-position HCS
plate (``:58-96``).
"""

import json

import numpy as np
import pytest
import yaml

from shrimpy_amd.io.omezarr import UnsupportedCodec, open_ome_zarr

N_T, N_C, N_Z, N_Y, N_X = 2, 2, 40, 6, 9
CHANNELS = ["BF", "GFP"]
KEYS = ["0/0/000", "0/1/000"]


def _encoded(p):
    d = np.zeros((N_T, N_C, N_Z, N_Y, N_X), dtype=np.uint16)
    for t in range(N_T):
        for c in range(N_C):
            for z in range(N_Z):
                d[t, c, z] = p * 30000 + t * 10000 + c * 1000 + z
    return d


def _make_plate(path, version, compress=None, dtype=np.uint16):
    with open_ome_zarr(path, layout="hcs", mode="w", channel_names=CHANNELS, version=version,
                       prefer_iohub=False) as plate:
        for p, key in enumerate(KEYS):
            pos = plate.create_position(*key.split("/"))
            arr = pos.create_zeros("0", shape=(N_T, N_C, N_Z, N_Y, N_X), dtype=dtype,
                                   scale=(1, 1, 0.15, 0.1133, 0.1133), compress=compress)
            data = _encoded(p).astype(dtype)
            for t in range(N_T):
                for c in range(N_C):
                    arr.write_volume(t, c, data[t, c])
    return path


@pytest.mark.parametrize("version", ["0.4", "0.5"])
@pytest.mark.parametrize("compress", [None, "zlib"])
def test_plate_roundtrip(tmp_path, version, compress):
    path = _make_plate(tmp_path / "plate.zarr", version, compress)
    with open_ome_zarr(path, layout="auto", mode="r", prefer_iohub=False) as plate:
        positions = dict(plate.positions())
        assert list(positions) == KEYS                      # "row/col/fov" keys, creation order
        for p, key in enumerate(KEYS):
            pos = positions[key]
            arr = pos["0"]
            assert arr.shape == (N_T, N_C, N_Z, N_Y, N_X) and arr.dtype == np.uint16
            assert arr.chunks == (1, 1, 32, N_Y, N_X)       # min(32, nz) z-chunks, whole planes
            np.testing.assert_array_equal(arr[:], _encoded(p))
            np.testing.assert_array_equal(arr.read_volume(1, 0), _encoded(p)[1, 0])
            assert pos.channel_names == CHANNELS
            assert pos.scale[2] == pytest.approx(0.15)      # index 2 = Z (replay_camera.py:258-268)


def test_store_metadata_on_disk(tmp_path):
    p4 = _make_plate(tmp_path / "v4.zarr", "0.4")
    assert json.loads((p4 / ".zgroup").read_text())["zarr_format"] == 2
    attrs = json.loads((p4 / ".zattrs").read_text())
    assert [w["path"] for w in attrs["plate"]["wells"]] == ["0/0", "0/1"]
    img = json.loads((p4 / "0" / "0" / "000" / ".zattrs").read_text())
    assert [a["name"] for a in img["multiscales"][0]["axes"]] == ["T", "C", "Z", "Y", "X"]
    p5 = _make_plate(tmp_path / "v5.zarr", "0.5")
    root = json.loads((p5 / "zarr.json").read_text())
    assert root["zarr_format"] == 3 and root["attributes"]["ome"]["version"] == "0.5"
    arr = json.loads((p5 / "0" / "0" / "000" / "0" / "zarr.json").read_text())
    assert arr["node_type"] == "array" and arr["shape"] == [N_T, N_C, N_Z, N_Y, N_X]


def test_fov_store_and_missing_chunks_read_as_fill(tmp_path):
    path = tmp_path / "fov.zarr"
    with open_ome_zarr(path, layout="fov", mode="w", channel_names=["BF"], prefer_iohub=False) as fov:
        arr = fov.create_zeros("0", shape=(1, 1, 5, 4, 3), dtype="float32")
        arr.write_volume(0, 0, np.arange(60, dtype=np.float32).reshape(5, 4, 3))
    with open_ome_zarr(path, prefer_iohub=False) as fov:
        assert [k for k, _ in fov.positions()] == ["0/0/0"]
        np.testing.assert_array_equal(fov["0"].read_volume(0, 0).ravel(), np.arange(60))
    # an all-zero (never written) volume, e.g. an autofocus-failed stack, reads as zeros
    path2 = tmp_path / "empty.zarr"
    with open_ome_zarr(path2, layout="fov", mode="w", prefer_iohub=False) as fov:
        fov.create_zeros("0", shape=(1, 1, 5, 4, 3), dtype="uint16")
    with open_ome_zarr(path2, prefer_iohub=False) as fov:
        assert not fov["0"].read_volume(0, 0).any()


def test_store_errors(tmp_path):
    path = _make_plate(tmp_path / "plate.zarr", "0.4")
    with pytest.raises(FileExistsError):  # never overwrite, like the reference (mantis_engine.py:458)
        open_ome_zarr(path, layout="hcs", mode="w", prefer_iohub=False)
    with pytest.raises(FileNotFoundError):
        open_ome_zarr(tmp_path / "nope.zarr", prefer_iohub=False)
    with pytest.raises(ValueError):
        open_ome_zarr(path, layout="fov", prefer_iohub=False)
    with open_ome_zarr(path, prefer_iohub=False) as plate:
        arr = dict(plate.positions())[KEYS[0]]["0"]
        with pytest.raises(PermissionError):
            arr.write_volume(0, 0, np.zeros((N_Z, N_Y, N_X)))
        with pytest.raises(IndexError):
            arr.read_volume(5, 0)
    # an array with a codec this reader does not implement is reported (with the fix), not mis-read
    meta = path / "0" / "0" / "000" / "0" / ".zarray"
    m = json.loads(meta.read_text())
    m["compressor"] = {"id": "lzma"}
    meta.write_text(json.dumps(m))
    with open_ome_zarr(path, prefer_iohub=False) as plate, pytest.raises(UnsupportedCodec, match="install iohub"):
        dict(plate.positions())[KEYS[0]]["0"]


# ------------------------------------------------------------------ CLI wiring (compute stubbed)


class _FakeReconstructor:
    """Stands in for VolumeReconstructor on the CPU: halves Z, keeps values recognisable."""

    def __init__(self, raw_shape, settings, device):
        self.raw_shape = raw_shape
        self.output_shape = (raw_shape[0] // 2, raw_shape[1], raw_shape[2])
        self.settings = settings

    def __call__(self, raw):
        import torch

        return torch.as_tensor(np.asarray(raw, dtype=np.float32)[::2] + 0.5)


@pytest.fixture
def cpu_cli(monkeypatch):
    import torch

    import shrimpy_amd.cli as cli

    monkeypatch.setattr(cli, "_distributed", lambda: (0, 1, torch.device("cpu")))
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    return cli


def test_cli_deskew_runs_every_unit_and_writes_scaled_store(tmp_path, cpu_cli):
    src = _make_plate(tmp_path / "raw.zarr", "0.4")
    cfg = tmp_path / "deskew.yml"
    cfg.write_text(yaml.safe_dump(dict(pixel_size_um=0.1133, ls_angle_deg=30.0, scan_step_um=0.15,
                                       keep_overhang=True, average_n_slices=3)))
    from shrimpy_amd.settings import DeskewSettings, ReconstructSettings

    out = tmp_path / "out.zarr"
    res = cpu_cli.run_store(src, out, ReconstructSettings(deskew=DeskewSettings.from_yaml(cfg)),
                            zarr_version="0.5", reconstructor_factory=_FakeReconstructor)
    assert res["units"] == res["units_total"] == len(KEYS) * N_T * N_C
    with open_ome_zarr(out, prefer_iohub=False) as plate:
        positions = dict(plate.positions())
        assert list(positions) == KEYS
        for p, key in enumerate(KEYS):
            arr = positions[key]["0"]
            assert arr.shape == (N_T, N_C, N_Z // 2, N_Y, N_X) and arr.dtype == np.float32
            np.testing.assert_array_equal(arr.read_volume(1, 1), _encoded(p)[1, 1][::2] + 0.5)
            # deskewed voxel size: (avg * sin(theta) * px, px, px)
            assert positions[key].scale[2:] == pytest.approx((3 * 0.5 * 0.1133, 0.1133, 0.1133))
            assert positions[key].channel_names == CHANNELS


def test_cli_commands_parse_configs_and_positions(tmp_path, cpu_cli, monkeypatch):
    from click.testing import CliRunner

    src = _make_plate(tmp_path / "raw.zarr", "0.4")
    seen = {}

    def fake_run_store(input_path, output_path, settings, positions, zarr_version):
        seen.update(settings=settings, positions=positions, version=zarr_version)
        return {"ok": True}

    monkeypatch.setattr(cpu_cli, "run_store", fake_run_store)
    runner = CliRunner()

    reg = tmp_path / "register.yml"
    reg.write_text(yaml.safe_dump(dict(affine_transform_zyx=np.eye(4).tolist())))
    r = runner.invoke(cpu_cli.cli, ["register", "-i", str(src), "-c", str(reg), "-o", str(tmp_path / "o1"),
                                    "-p", KEYS[1]])
    assert r.exit_code == 0, r.output
    assert seen["settings"].registration is not None and seen["positions"] == (KEYS[1],)

    dec = tmp_path / "deconvolve.yml"
    dec.write_text(yaml.safe_dump(dict(iterations=7)))
    r = runner.invoke(cpu_cli.cli, ["deconvolve", "-i", str(src), "-c", str(dec), "-o", str(tmp_path / "o2"),
                                    "--zarr-version", "0.5"])
    assert r.exit_code == 0, r.output
    assert seen["settings"].deconvolution.iterations == 7 and seen["version"] == "0.5"

    bad = tmp_path / "bad.yml"
    bad.write_text(yaml.safe_dump(dict(iterations=7, bogus=1)))  # extra="forbid"
    r = runner.invoke(cpu_cli.cli, ["deconvolve", "-i", str(src), "-c", str(bad), "-o", str(tmp_path / "o3")])
    assert r.exit_code != 0

    r = runner.invoke(cpu_cli.cli, ["-h"])
    assert r.exit_code == 0 and all(c in r.output for c in ("deskew", "register", "deconvolve", "reconstruct"))


def test_cli_unknown_position_is_an_error(tmp_path, cpu_cli):
    import click

    src = _make_plate(tmp_path / "raw.zarr", "0.4")
    from shrimpy_amd.settings import DeconvolveSettings, ReconstructSettings

    with pytest.raises(click.ClickException, match="not found"):
        cpu_cli.run_store(src, tmp_path / "o", ReconstructSettings(deconvolution=DeconvolveSettings()),
                          positions=("9/9/9",), reconstructor_factory=_FakeReconstructor)


@pytest.mark.gpu
def test_cli_reconstruct_end_to_end_on_gpu(tmp_path):
    """Real kernels through the CLI: tiny plate in, deskew + 3 RL iterations, store out; one volume
    compared with the oracle."""
    from click.testing import CliRunner

    from oracle import cpu_ref as o
    from shrimpy_amd.cli import cli

    rng = np.random.default_rng(5)
    path = tmp_path / "raw.zarr"
    vols = {}
    with open_ome_zarr(path, layout="hcs", mode="w", channel_names=["BF"], prefer_iohub=False) as plate:
        for key in KEYS:
            pos = plate.create_position(*key.split("/"))
            arr = pos.create_zeros("0", shape=(1, 1, 64, 16, 40), dtype="uint16", scale=(1, 1, 0.15, 0.1133, 0.1133))
            vols[key] = rng.integers(80, 600, (64, 16, 40)).astype(np.uint16)
            arr.write_volume(0, 0, vols[key])
    cfg = tmp_path / "recon.yml"
    cfg.write_text(yaml.safe_dump(dict(
        deskew=dict(pixel_size_um=0.1133, ls_angle_deg=30.0, scan_step_um=0.15, keep_overhang=True,
                    average_n_slices=3),
        deconvolution=dict(iterations=3, gaussian_shape_zyx=[5, 5, 5], gaussian_sigma_zyx=[1.2, 1.0, 1.0]))))
    out = tmp_path / "recon.zarr"
    r = CliRunner().invoke(cli, ["reconstruct", "-i", str(path), "-c", str(cfg), "-o", str(out)])
    assert r.exit_code == 0, r.output
    psf, _ = o.gaussian_psf((5, 5, 5), (1.2, 1.0, 1.0))
    with open_ome_zarr(out, prefer_iohub=False) as plate:
        for key, pos in plate.positions():
            got = pos["0"].read_volume(0, 0).astype(np.float64)
            d = o.deskew(vols[key].astype(np.float32), 30.0, 0.755, True, 3)
            ref = o.richardson_lucy(d, psf, 3).astype(np.float64)
            assert got.shape == ref.shape
            assert np.all(np.abs(got - ref) <= 5e-5 * np.abs(ref) + 2e-5 * np.abs(ref).max())


@pytest.mark.gpu
@pytest.mark.parametrize("version", ["0.4", "0.5"])
def test_cli_deskew_register_deconvolve_chain_on_gpu(tmp_path, version):
    """The three single-step commands chained store to store (two timepoints, two channels), each
    stage compared with the oracle: deskew and register bit for bit, deconvolve within the RL bar."""
    from click.testing import CliRunner

    from oracle import cpu_ref as o
    from shrimpy_amd.cli import cli

    rng = np.random.default_rng(8)
    raw = tmp_path / "raw.zarr"
    vols = {}
    with open_ome_zarr(raw, layout="hcs", mode="w", channel_names=["LS1", "LS2"], version=version,
                       prefer_iohub=False) as plate:
        for key in KEYS:
            pos = plate.create_position(*key.split("/"))
            arr = pos.create_zeros("0", shape=(2, 2, 48, 12, 34), dtype="uint16", scale=(1, 1, 0.15, 0.1133, 0.1133))
            for t in range(2):
                for c in range(2):
                    vols[key, t, c] = rng.integers(80, 900, (48, 12, 34)).astype(np.uint16)
                    arr.write_volume(t, c, vols[key, t, c])
    th = np.deg2rad(3.0)
    m = np.eye(4)
    m[:3, :3] = np.array([[1, 0, 0], [0, np.cos(th), -np.sin(th)], [0, np.sin(th), np.cos(th)]]) @ np.diag([1.0, 0.97, 1.03])
    m[:3, 3] = [0.5, -2.25, 3.75]
    (tmp_path / "deskew.yml").write_text(yaml.safe_dump(dict(
        pixel_size_um=0.1133, ls_angle_deg=30.0, scan_step_um=0.15, keep_overhang=False, average_n_slices=2)))
    (tmp_path / "register.yml").write_text(yaml.safe_dump(dict(affine_transform_zyx=m.tolist(), cval=0.0)))
    (tmp_path / "deconvolve.yml").write_text(yaml.safe_dump(dict(
        iterations=4, gaussian_shape_zyx=[3, 5, 5], gaussian_sigma_zyx=[0.8, 1.0, 1.0])))
    stages = [("deskew", raw, tmp_path / "d.zarr"), ("register", tmp_path / "d.zarr", tmp_path / "r.zarr"),
              ("deconvolve", tmp_path / "r.zarr", tmp_path / "x.zarr")]
    for cmd, src, dst in stages:
        r = CliRunner().invoke(cli, [cmd, "-i", str(src), "-c", str(tmp_path / f"{cmd}.yml"), "-o", str(dst),
                                     "--zarr-version", version])
        assert r.exit_code == 0, (cmd, r.output, r.exception)
    psf, _ = o.gaussian_psf((3, 5, 5), (0.8, 1.0, 1.0))
    stores = {name: open_ome_zarr(path, prefer_iohub=False) for name, _, path in stages}
    try:
        pos = {name: dict(st.positions()) for name, st in stores.items()}
        for (key, t, c), v in vols.items():
            d = o.deskew(v.astype(np.float32), 30.0, 0.755, False, 2)
            np.testing.assert_array_equal(pos["deskew"][key]["0"].read_volume(t, c), d)
            reg = o.affine_apply_4x4(d, m, d.shape)
            np.testing.assert_array_equal(pos["register"][key]["0"].read_volume(t, c), reg)
            ref = o.richardson_lucy(reg, psf, 4).astype(np.float64)
            got = pos["deconvolve"][key]["0"].read_volume(t, c).astype(np.float64)
            assert np.all(np.abs(got - ref) <= 2e-4 * np.abs(ref) + 1e-4 * np.abs(ref).max())
    finally:
        for st in stores.values():
            st.close()
