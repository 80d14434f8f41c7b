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

    monkeypatch.setattr(cli, "_distributed", lambda: (0, 1, torch.device("cpu"), False))
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

    def fake_run_store(input_path, output_path, settings, positions, zarr_version, resume=False,
                       io_backend="auto", compression=None, on_error="raise"):
        seen.update(settings=settings, positions=positions, version=zarr_version, resume=resume,
                    io_backend=io_backend, compression=compression, on_error=on_error)
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
    assert (seen["resume"], seen["io_backend"], seen["compression"]) == (False, "auto", "blosc-zstd")   # the reference's format
    r = runner.invoke(cpu_cli.cli, ["deconvolve", "-i", str(src), "-c", str(dec), "-o", str(tmp_path / "o2"),
                                    "--resume", "--io", "native", "--compression", "blosc-zstd"])
    assert r.exit_code == 0, r.output
    assert (seen["resume"], seen["io_backend"], seen["compression"], seen["on_error"]) == (True, "native", "blosc-zstd", "raise")
    r = runner.invoke(cpu_cli.cli, ["deconvolve", "-i", str(src), "-c", str(dec), "-o", str(tmp_path / "o3"), "--on-error", "skip"])
    assert r.exit_code == 0 and seen["on_error"] == "skip"

    # [RECALLED] biahub spellings: register -s (moving) / -t (target: its (Z, Y, X) is the output shape when the config
    # names none); deconvolve --psf-dirpath overrides the config's psf_path
    tgt = tmp_path / "target.zarr"
    with open_ome_zarr(tgt, layout="hcs", mode="w", channel_names=["LF"], prefer_iohub=False) as plate:
        plate.create_position("A", "1", "0").create_zeros("0", shape=(1, 1, 9, 21, 33), dtype="float32")
    r = runner.invoke(cpu_cli.cli, ["register", "-s", str(src), "-t", str(tgt), "-c", str(reg), "-o", str(tmp_path / "o4")])
    assert r.exit_code == 0, r.output
    assert tuple(seen["settings"].registration.output_shape_zyx) == (9, 21, 33)
    pinned = tmp_path / "register_shape.yml"
    pinned.write_text(yaml.safe_dump(dict(affine_transform_zyx=np.eye(4).tolist(), output_shape_zyx=[4, 5, 6])))
    r = runner.invoke(cpu_cli.cli, ["register", "-s", str(src), "-t", str(tgt), "-c", str(pinned), "-o", str(tmp_path / "o5")])
    assert r.exit_code == 0 and tuple(seen["settings"].registration.output_shape_zyx) == (4, 5, 6)      # the config wins
    r = runner.invoke(cpu_cli.cli, ["register", "-i", str(src), "-s", str(src), "-c", str(reg), "-o", str(tmp_path / "o6")])
    assert r.exit_code != 0 and "-i or -s" in r.output
    r = runner.invoke(cpu_cli.cli, ["register", "-c", str(reg), "-o", str(tmp_path / "o6")])
    assert r.exit_code != 0 and "-i or -s" in r.output
    np.save(tmp_path / "beads.npy", np.ones((3, 3, 3), np.float32) / 27)
    r = runner.invoke(cpu_cli.cli, ["deconvolve", "-i", str(src), "-c", str(dec), "-o", str(tmp_path / "o7"),
                                    "--psf-dirpath", str(tmp_path / "beads.npy")])
    assert r.exit_code == 0, r.output
    assert seen["settings"].deconvolution.psf_path == str(tmp_path / "beads.npy") and seen["settings"].deconvolution.iterations == 7

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


# ------------------------------------------------------------------ resume (streamed runs, config 5)


def _settings():
    from shrimpy_amd.settings import DeconvolveSettings, ReconstructSettings

    return ReconstructSettings(deconvolution=DeconvolveSettings(iterations=3))


def _read_all(path):
    with open_ome_zarr(path, prefer_iohub=False) as plate:
        return {k: p["0"][:] for k, p in plate.positions()}


@pytest.mark.parametrize("version", ["0.4", "0.5"])
def test_resume_after_a_failed_store_equals_an_uninterrupted_run(tmp_path, cpu_cli, monkeypatch, version):
    """A run dies in ``store`` after k of n units (disk full, node lost ...).  ``--resume`` skips the
    units recorded as complete, rewrites the rest -- the half-written one included -- and the store
    then equals the one an uninterrupted run writes.  Unit = one (position, t, c) volume, the
    reference's own granularity (``mantis_engine.py:458, 480``: never overwrite; ``tracking.py:887-914``:
    row-by-row append)."""
    import click

    from shrimpy_amd.io.omezarr import ZarrArray

    src = _make_plate(tmp_path / "raw.zarr", version)
    n = len(KEYS) * N_T * N_C
    want = tmp_path / "whole.zarr"
    cpu_cli.run_store(src, want, _settings(), zarr_version=version, reconstructor_factory=_FakeReconstructor)

    out = tmp_path / "out.zarr"
    real_write = ZarrArray.write_volume
    written = []

    def flaky_write(self, *args):
        if len(written) == 5:
            # the sixth unit: leave a partial volume behind, then die
            half = np.array(args[-1], copy=True)
            half[half.shape[0] // 2:] = -1
            real_write(self, *args[:-1], half)
            raise OSError("No space left on device")
        real_write(self, *args)
        written.append(args[:-1])

    monkeypatch.setattr(ZarrArray, "write_volume", flaky_write)
    with pytest.raises(OSError, match="No space"):
        cpu_cli.run_store(src, out, _settings(), zarr_version=version, reconstructor_factory=_FakeReconstructor)
    monkeypatch.setattr(ZarrArray, "write_volume", real_write)
    assert len(written) == 5
    assert any(not np.array_equal(a, b) for a, b in zip(_read_all(out).values(), _read_all(want).values()))

    with pytest.raises(click.ClickException, match="--resume"):       # without the flag: never overwrite
        cpu_cli.run_store(src, out, _settings(), zarr_version=version, reconstructor_factory=_FakeReconstructor)

    calls = []

    class Counting(_FakeReconstructor):
        def __call__(self, raw):
            calls.append(1)
            return super().__call__(raw)

    res = cpu_cli.run_store(src, out, _settings(), zarr_version=version, reconstructor_factory=Counting,
                            resume=True)
    assert res["units_skipped"] == 5 and res["units"] == n - 5 == len(calls)
    got, ref = _read_all(out), _read_all(want)
    assert list(got) == list(ref)
    for k in ref:
        np.testing.assert_array_equal(got[k], ref[k])
    # a second --resume has nothing left to do
    res = cpu_cli.run_store(src, out, _settings(), zarr_version=version, reconstructor_factory=Counting, resume=True)
    assert res["units_skipped"] == n and res["units"] == 0


def test_on_error_skip_leaves_out_the_unit_with_a_corrupt_chunk_and_resume_retries_only_it(tmp_path, cpu_cli):
    """Round-4 verdict: one damaged chunk ended a whole run.  With ``on_error="skip"`` the unit whose chunk does not decode
    is listed (result, ``.lsr_failed/``), every other unit is written, the CLI exits 3; after the chunk is repaired
    ``--resume`` reconstructs that one unit only.  Reference: a failed stack becomes an error record and the acquisition
    carries on (``shrimpy/dynatrack/worker.py:262-271``)."""
    import json

    from click.testing import CliRunner

    src = _make_plate(tmp_path / "raw.zarr", "0.5", compress="zlib")
    n = len(KEYS) * N_T * N_C
    want = tmp_path / "whole.zarr"
    cpu_cli.run_store(src, want, _settings(), zarr_version="0.5", reconstructor_factory=_FakeReconstructor)
    # damage one chunk of position 1, (t, c) = (1, 0)
    victim = next(f for f in sorted((src / "0" / "1" / "000" / "0" / "c" / "1" / "0").rglob("*")) if f.is_file())
    good = victim.read_bytes()
    victim.write_bytes(good[:len(good) // 2])
    out = tmp_path / "out.zarr"
    with pytest.raises(Exception):                                     # the default: the run stops at the damaged unit
        cpu_cli.run_store(src, tmp_path / "stops.zarr", _settings(), zarr_version="0.5", reconstructor_factory=_FakeReconstructor)
    res = cpu_cli.run_store(src, out, _settings(), zarr_version="0.5", reconstructor_factory=_FakeReconstructor, on_error="skip")
    assert [(f["position"], f["t"], f["c"], f["stage"]) for f in res["failed"]] == [(KEYS[1], 1, 0, "load")]
    record = json.loads((out / ".lsr_failed" / KEYS[1].replace("/", "__") / "t1_c0.json").read_text())
    assert record["stage"] == "load" and record["error"]
    got, ref = _read_all(out), _read_all(want)
    bad_key = next(k for k in ref if not np.array_equal(got[k], ref[k]))
    assert all(np.array_equal(got[k], ref[k]) for k in ref if k != bad_key)    # every other unit is there
    # the same through the command line: exit status 3, the failure on stderr
    cfg = tmp_path / "cfg.yaml"
    cfg.write_text("deskew:\n  pixel_size_um: 0.1133\n  scan_step_um: 0.15\n  ls_angle_deg: 30.0\n  average_n_slices: 1\n")
    run = CliRunner().invoke(cpu_cli.cli, ["reconstruct", "-i", str(src), "-c", str(cfg), "-o", str(tmp_path / "cli.zarr"),
                                           "--on-error", "skip", "--compression", "none"])
    assert run.exit_code == 3 and "FAILED" in run.output and KEYS[1] in run.output
    # repaired: --resume does the one unit, and the failure record goes away
    victim.write_bytes(good)
    calls = []

    class Counting(_FakeReconstructor):
        def __call__(self, raw):
            calls.append(1)
            return super().__call__(raw)

    res = cpu_cli.run_store(src, out, _settings(), zarr_version="0.5", reconstructor_factory=Counting, resume=True, on_error="skip")
    assert res["failed"] == [] and res["units_skipped"] == n - 1 and len(calls) == 1
    assert not list((out / ".lsr_failed").glob("*/*.json"))
    got = _read_all(out)
    for k in ref:
        np.testing.assert_array_equal(got[k], ref[k])


def test_resume_refuses_a_store_written_with_other_settings(tmp_path, cpu_cli):
    import click

    from shrimpy_amd.settings import DeconvolveSettings, ReconstructSettings

    src = _make_plate(tmp_path / "raw.zarr", "0.4")
    out = tmp_path / "out.zarr"
    cpu_cli.run_store(src, out, _settings(), reconstructor_factory=_FakeReconstructor)
    other = ReconstructSettings(deconvolution=DeconvolveSettings(iterations=4))
    with pytest.raises(click.ClickException, match="different settings"):
        cpu_cli.run_store(src, out, other, reconstructor_factory=_FakeReconstructor, resume=True)
    # --resume on a fresh path is simply a fresh run
    res = cpu_cli.run_store(src, tmp_path / "new.zarr", _settings(), reconstructor_factory=_FakeReconstructor,
                            resume=True)
    assert res["units_skipped"] == 0 and res["units"] == res["units_total"]


def test_resume_with_a_subset_of_positions_then_the_rest(tmp_path, cpu_cli):
    """``-p`` runs are their own run (the fingerprint covers the positions): a store holding one
    position is not silently extended by a run over different ones."""
    import click

    src = _make_plate(tmp_path / "raw.zarr", "0.4")
    out = tmp_path / "out.zarr"
    cpu_cli.run_store(src, out, _settings(), positions=(KEYS[0],), reconstructor_factory=_FakeReconstructor)
    with pytest.raises(click.ClickException, match="different input or with different settings"):
        cpu_cli.run_store(src, out, _settings(), reconstructor_factory=_FakeReconstructor, resume=True)


def test_staged_run_drains_the_stager_when_a_loader_fails():
    """``_run_staged`` with a loader that raises mid-run: the exception propagates, nothing is left
    in flight (``drain`` ran, no worker thread is still filling a slot), finished units stay stored."""
    import threading

    from shrimpy_amd.pipeline import run_sharded

    class Stager:
        depth = 2

        def __init__(self):
            # raw slots and result slots are separate buffers, as in VolumeStager: the loader fills the
            # raw slot of unit i + 1 while the writer still reads the result slot of unit i - 1
            self.events, self.slots = [], [np.zeros(4, np.float32) for _ in range(2)]
            self.results = [None, None]

        def host_in(self, slot):
            return self.slots[slot]

        def stage_in(self, slot, data):
            self.events.append(("in", slot))
            return slot

        def acquire(self, slot):
            return self.slots[slot].copy()

        def release(self, slot):
            pass

        def stage_out(self, slot, result):
            self.results[slot] = np.asarray(result)

        def collect(self, slot):
            return self.results[slot]

        def drain(self):
            self.events.append("drain")

    for fail_in in ("load", "process", "store"):
        st, stored = Stager(), []

        def load(u, out=None):
            if fail_in == "load" and u == 3:
                raise OSError("chunk vanished")
            out[:] = u
            return out

        def process(v):
            if fail_in == "process" and int(v[0]) == 3:
                raise RuntimeError("kernel failed")
            return v + 100

        def store(u, vol):
            if fail_in == "store" and u == 3:
                raise OSError("disk full")
            stored.append((u, float(vol[0])))

        before = threading.active_count()
        with pytest.raises((OSError, RuntimeError)):
            run_sharded(list(range(6)), load, process, store, stager=st)
        assert st.events[-1] == "drain"
        assert threading.active_count() == before
        assert stored == [(0, 100.0), (1, 101.0), (2, 102.0)]


# ------------------------------------------------------------------ channels, plates, backends


def test_register_warps_only_the_source_channels(tmp_path, cpu_cli):
    """``source_channel_names`` / ``target_channel_name`` are honoured: the named sources go through
    the registration, every other channel (the target) through the same pipeline without it."""
    import click

    from shrimpy_amd.settings import ReconstructSettings, RegisterSettings

    src = _make_plate(tmp_path / "raw.zarr", "0.4")
    seen = []

    class Rec:
        def __init__(self, raw_shape, settings, device):
            self.output_shape = tuple(raw_shape)
            self.warps = settings.registration is not None

        def __call__(self, raw):
            import torch

            seen.append(self.warps)
            return torch.as_tensor(np.asarray(raw, dtype=np.float32) + (1000.0 if self.warps else 0.0))

    eye = np.eye(4).tolist()
    s = ReconstructSettings(registration=RegisterSettings(
        affine_transform_zyx=eye, source_channel_names=[CHANNELS[1]], target_channel_name=CHANNELS[0]))
    cpu_cli.run_store(src, tmp_path / "o.zarr", s, reconstructor_factory=Rec)
    got = _read_all(tmp_path / "o.zarr")
    for p, key in enumerate(KEYS):
        np.testing.assert_array_equal(got[key][:, 0], _encoded(p)[:, 0].astype(np.float32))           # target: as is
        np.testing.assert_array_equal(got[key][:, 1], _encoded(p)[:, 1].astype(np.float32) + 1000.0)  # source: warped
    assert sum(seen) == len(seen) // 2

    for bad, msg in [(dict(source_channel_names=["nope"]), "not in the store"),
                     (dict(source_channel_names=[CHANNELS[0]], target_channel_name=CHANNELS[0]), "also listed"),
                     (dict(target_channel_name=CHANNELS[0]), "no source_channel_names")]:
        s = ReconstructSettings(registration=RegisterSettings(affine_transform_zyx=eye, **bad))
        with pytest.raises(click.ClickException, match=msg):
            cpu_cli.run_store(src, tmp_path / "bad.zarr", s, reconstructor_factory=Rec)
        assert not (tmp_path / "bad.zarr").exists()
    # a changed output shape cannot be mixed with unwarped channels
    s = ReconstructSettings(registration=RegisterSettings(
        affine_transform_zyx=eye, source_channel_names=[CHANNELS[1]], output_shape_zyx=(4, 4, 4)))

    class Reshaping(Rec):
        def __init__(self, raw_shape, settings, device):
            super().__init__(raw_shape, settings, device)
            if settings.registration is not None:
                self.output_shape = (4, 4, 4)

    with pytest.raises(click.ClickException, match="output_shape_zyx"):
        cpu_cli.run_store(src, tmp_path / "bad.zarr", s, reconstructor_factory=Reshaping)


def test_input_option_takes_one_store_or_the_position_directories_a_glob_expands_to(tmp_path, cpu_cli):
    """[RECALLED] biahub ``-i/--input-position-dirpaths`` eats every following path (``plate.zarr/*/*/*``); both forms
    select the same units, position directories are grouped by their plate, mixed plates are refused, ``-p`` narrows."""
    import click

    from click.testing import CliRunner

    src = _make_plate(tmp_path / "raw.zarr", "0.5")
    other = _make_plate(tmp_path / "other.zarr", "0.5")
    assert cpu_cli.resolve_inputs([src]) == (src, ())
    dirs = [src / k for k in KEYS]
    assert cpu_cli.resolve_inputs(dirs) == (src.resolve(), tuple(KEYS))
    assert cpu_cli.resolve_inputs([dirs[1]]) == (src.resolve(), (KEYS[1],))      # one position of a plate
    for bad, why in (([dirs[0], other / KEYS[1]], "different plates"), ([dirs[0], dirs[0]], "twice"),
                     ([tmp_path / "nope"], "does not exist"), ([src, other], "position director")):
        with pytest.raises(click.ClickException, match=why):
            cpu_cli.resolve_inputs(bad)
    cfg = tmp_path / "deskew.yml"
    cfg.write_text(yaml.safe_dump(dict(pixel_size_um=0.1133, ls_angle_deg=30.0, scan_step_um=0.15,
                                       keep_overhang=True, average_n_slices=3)))
    seen = []

    def fake_run_store(input_path, output_path, settings, positions=(), zarr_version="0.5", **kw):
        seen.append((input_path, tuple(positions), zarr_version, kw.get("compression")))
        return {}

    import shrimpy_amd.cli as real

    real_run = real.run_store
    real.run_store = fake_run_store
    try:
        run = CliRunner()
        # the glob form: -i eats the paths up to the next option
        r = run.invoke(real.cli, ["deskew", "-i", str(dirs[0]), str(dirs[1]), "-c", str(cfg), "-o", str(tmp_path / "a.zarr")])
        assert r.exit_code == 0, r.output
        r = run.invoke(real.cli, ["deskew", "-c", str(cfg), "-o", str(tmp_path / "b.zarr"), "-i", str(src)])
        assert r.exit_code == 0, r.output
        r = run.invoke(real.cli, ["deskew", "-i", str(dirs[0]), str(dirs[1]), "-p", KEYS[1], "-c", str(cfg), "-o", str(tmp_path / "c.zarr")])
        assert r.exit_code == 0, r.output
        r = run.invoke(real.cli, ["deskew", "-i", str(dirs[0]), "-p", KEYS[1], "-c", str(cfg), "-o", str(tmp_path / "d.zarr")])
        assert r.exit_code != 0 and "not among the position directories" in r.output
    finally:
        real.run_store = real_run
    assert seen[0][:2] == (src.resolve(), tuple(KEYS)) and seen[1][:2] == (src, ()) and seen[2][:2] == (src.resolve(), (KEYS[1],))
    # the output defaults are the reference's own format: NGFF 0.5 (tracking.py:1337-1343), blosc-zstd (mantis_engine.py:474-481)
    assert seen[0][2:] == ("0.5", "blosc-zstd")


def test_registration_keep_overhang_grows_the_output_to_the_union_box(tmp_path):
    """[RECALLED] biahub RegisterSettings.keep_overhang: the output covers the target grid AND the moving volume's
    footprint; the box's corner is folded into the matrix.  Values against the oracle's affine apply on that box
    (CPU tensors: the host twin, bit-equal to scipy)."""
    import torch

    from oracle import cpu_ref as o
    from shrimpy_amd.pipeline import VolumeReconstructor
    from shrimpy_amd.settings import ReconstructSettings, RegisterSettings

    rng = np.random.default_rng(3)
    vol = rng.random((6, 10, 14)).astype(np.float32)
    th = np.deg2rad(10.0)
    m = np.eye(4)
    m[1:3, 1:3] = [[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]
    m[:3, 3] = [1.5, -2.0, 3.25]
    plain = RegisterSettings(affine_transform_zyx=m.tolist())
    assert plain.resolved(vol.shape)[1:] == (vol.shape, (0, 0, 0))
    grown = RegisterSettings(affine_transform_zyx=m.tolist(), keep_overhang=True)
    mat, shape, origin = grown.resolved(vol.shape)
    # every corner of the moving volume lands inside the box, and so does the target grid
    corners = np.array([[z, y, x] for z in (0, 5) for y in (0, 9) for x in (0, 13)], float)
    in_target = (np.linalg.inv(m[:3, :3]) @ (corners - m[:3, 3]).T).T - np.asarray(origin)
    assert np.all(in_target >= -1e-9) and np.all(in_target <= np.asarray(shape) - 1 + 1e-9)
    assert all(o_ <= 0 for o_ in origin) and all(o_ + n >= t for o_, n, t in zip(origin, shape, vol.shape))
    assert shape != vol.shape
    rec = VolumeReconstructor(vol.shape, ReconstructSettings(registration=grown), "cpu")
    assert rec.output_shape == shape and rec.register_origin == origin
    got = rec(torch.as_tensor(vol)).numpy()
    np.testing.assert_array_equal(got, o.affine_apply_4x4(vol, mat, shape))
    # the target window of the grown output is the plain registration
    z0, y0, x0 = (-v for v in origin)
    want = VolumeReconstructor(vol.shape, ReconstructSettings(registration=plain), "cpu")(torch.as_tensor(vol)).numpy()
    np.testing.assert_array_equal(got[z0:z0 + 6, y0:y0 + 10, x0:x0 + 14], want)
    # the moving volume's content survives: its sum in the grown output is (close to) the whole of it, not a crop
    assert got.sum() > want.sum() * 1.05
    with pytest.raises(ValueError, match="invertible"):
        RegisterSettings(affine_transform_zyx=np.diag([1, 0, 1, 1.0]).tolist(), keep_overhang=True).resolved(vol.shape)


def test_keep_overhang_store_records_where_the_grown_grid_sits(tmp_path, cpu_cli):
    """``register_origin`` was computed and dropped -- the grown output's place in target coordinates was lost.
    It is now the NGFF ``translation`` (origin x scale, after the scale transform) of the output level; and a run that
    passes a channel through unwarped is told that keep_overhang is the reason its shapes cannot agree."""
    import click

    from shrimpy_amd.settings import ReconstructSettings, RegisterSettings

    src = _make_plate(tmp_path / "raw.zarr", "0.5", dtype=np.float32)
    m = np.eye(4)
    m[:3, 3] = [1.5, -2.0, 3.25]
    reg = RegisterSettings(affine_transform_zyx=m.tolist(), keep_overhang=True)
    _, shape, origin = reg.resolved((N_Z, N_Y, N_X))
    assert any(origin)
    res = cpu_cli.run_store(src, tmp_path / "out.zarr", ReconstructSettings(registration=reg), zarr_version="0.5")
    assert tuple(res["output_shape"]) == tuple(shape)
    meta = json.loads((tmp_path / "out.zarr" / "0" / "0" / "000" / "zarr.json").read_text())
    tr = meta["attributes"]["ome"]["multiscales"][0]["datasets"][0]["coordinateTransformations"]
    assert [t["type"] for t in tr] == ["scale", "translation"]
    scale = tr[0]["scale"]
    assert tr[1]["translation"] == [0.0, 0.0] + [float(o) * float(s) for o, s in zip(origin, scale[2:])]
    # without keep_overhang: no translation entry
    res = cpu_cli.run_store(src, tmp_path / "plain.zarr", ReconstructSettings(registration=RegisterSettings(affine_transform_zyx=m.tolist())),
                            zarr_version="0.5")
    meta = json.loads((tmp_path / "plain.zarr" / "0" / "0" / "000" / "zarr.json").read_text())
    assert [t["type"] for t in meta["attributes"]["ome"]["multiscales"][0]["datasets"][0]["coordinateTransformations"]] == ["scale"]
    partial = RegisterSettings(affine_transform_zyx=m.tolist(), keep_overhang=True, source_channel_names=[CHANNELS[0]])
    with pytest.raises(click.ClickException, match="keep_overhang"):
        cpu_cli.run_store(src, tmp_path / "no.zarr", ReconstructSettings(registration=partial), zarr_version="0.5")


def test_heterogeneous_plates_are_refused_before_anything_is_written(tmp_path, cpu_cli):
    import click

    path = tmp_path / "mixed.zarr"
    with open_ome_zarr(path, layout="hcs", mode="w", channel_names=["BF"], prefer_iohub=False) as plate:
        plate.create_position("A", "1", "0").create_zeros("0", shape=(1, 1, 8, 4, 6), dtype="uint16")
        plate.create_position("A", "2", "0").create_zeros("0", shape=(1, 1, 8, 4, 7), dtype="uint16")
        plate.create_position("A", "3", "0").create_zeros("0", shape=(1, 1, 8, 4, 6), dtype="float32")
    with pytest.raises(click.ClickException, match="differ from A/1/0"):
        cpu_cli.run_store(path, tmp_path / "o.zarr", _settings(), reconstructor_factory=_FakeReconstructor)
    assert not (tmp_path / "o.zarr").exists()
    res = cpu_cli.run_store(path, tmp_path / "o.zarr", _settings(), positions=("A/1/0",),
                            reconstructor_factory=_FakeReconstructor)
    assert res["units"] == 1


def test_cli_reads_the_engine_format_and_writes_it_back(tmp_path, cpu_cli):
    """Input in the acquisition's own layout (NGFF 0.5, sharded blosc-zstd, position ``A/1/fov0``,
    ``mantis_engine.py:474-481``) -> ``run_store`` -> output compressed the same way."""
    rng = np.random.default_rng(3)
    src = tmp_path / "acq.ome.zarr"
    vols = rng.integers(80, 600, (2, 1, 40, 12, 20)).astype(np.uint16)
    with open_ome_zarr(src, layout="hcs", mode="w", channel_names=["BF"], version="0.5", prefer_iohub=False) as plate:
        arr = plate.create_position("A", "1", "fov0").create_zeros(
            "0", shape=vols.shape, dtype="uint16", chunks=(1, 1, 16, 12, 20), compress="blosc-zstd", shards="volume")
        for t in range(2):
            arr.write_volume(t, 0, vols[t, 0])
    res = cpu_cli.run_store(src, tmp_path / "o.zarr", _settings(), zarr_version="0.5",
                            reconstructor_factory=_FakeReconstructor, compression="blosc-zstd", io_backend="native")
    assert res["units"] == 2
    with open_ome_zarr(tmp_path / "o.zarr", prefer_iohub=False) as plate:
        arr = plate["A/1/fov0"]["0"]
        assert arr._codec.kind == "blosc" and arr._codec.params["cname"] == "zstd"
        for t in range(2):
            np.testing.assert_array_equal(arr.read_volume(t, 0), vols[t, 0, ::2].astype(np.float32) + 0.5)


class _FakeIohub:
    """A stand-in ``iohub`` package with the surface the reference uses (``replay_camera.py:176-268``,
    ``tracking.py:1337-1367``, ``measure_psf.py:273-287``): ``open_ome_zarr(path, layout=, mode=
    [, channel_names=, version=])``, ``positions()`` -> ``(key, Position)``, ``Position["0"]`` /
    ``.data`` as an array with numpy indexing only (NO read_volume / write_volume),
    ``.channel_names``, ``.zattrs``, ``create_position``, ``create_zeros(name, shape, dtype, chunks,
    transform=[TransformationMeta])``.  Data lives in ``.npy`` files so that separate opens see it."""

    class TransformationMeta:
        def __init__(self, type, scale=None, translation=None):
            self.type, self.scale = type, scale

    class _Array:
        def __init__(self, path):
            self._path = path
            self._a = np.load(path, mmap_mode="r+")
            self.shape, self.dtype, self.chunks = self._a.shape, self._a.dtype, (1, 1) + self._a.shape[2:]

        def __getitem__(self, key):
            return np.array(self._a[key])

        def __setitem__(self, key, value):
            self._a[key] = value
            self._a.flush()

    class _Position:
        def __init__(self, root, key):
            self._dir = root / key
            self._meta = self._dir / "meta.json"

        @property
        def _m(self):
            return json.loads(self._meta.read_text())

        @property
        def channel_names(self):
            return self._m["channel_names"]

        @property
        def zattrs(self):
            return {"multiscales": [{"datasets": [{"coordinateTransformations": [
                {"type": "scale", "scale": self._m["scale"]}]}]}]}

        def __getitem__(self, name):
            return _FakeIohub._Array(self._dir / f"{name}.npy")

        @property
        def data(self):
            return self["0"]

        def create_zeros(self, name, shape, dtype, chunks=None, transform=None):
            np.save(self._dir / f"{name}.npy", np.zeros(shape, dtype))
            m = self._m
            m["scale"] = list(transform[0].scale) if transform else [1.0] * 5
            self._meta.write_text(json.dumps(m))
            return self[name]

    class _Plate:
        def __init__(self, path, mode, channel_names=None, version="0.4"):
            from pathlib import Path

            self.path, self.mode = Path(path), mode
            if mode == "w":
                if self.path.exists() and any(self.path.iterdir()):
                    raise FileExistsError(str(path))
                self.path.mkdir(parents=True, exist_ok=True)
                (self.path / "plate.json").write_text(json.dumps(
                    {"channel_names": list(channel_names or []), "version": version, "positions": []}))
            self.opened_with = dict(mode=mode, channel_names=channel_names, version=version)

        @property
        def _m(self):
            return json.loads((self.path / "plate.json").read_text())

        def positions(self):
            for key in self._m["positions"]:
                yield key, _FakeIohub._Position(self.path, key)

        def create_position(self, row, col, fov):
            key = f"{row}/{col}/{fov}"
            m = self._m
            m["positions"].append(key)
            (self.path / "plate.json").write_text(json.dumps(m))
            (self.path / key).mkdir(parents=True)
            (self.path / key / "meta.json").write_text(json.dumps(
                {"channel_names": m["channel_names"], "scale": [1.0] * 5}))
            return _FakeIohub._Position(self.path, key)

        def close(self):
            pass

    opened = []

    @classmethod
    def install(cls, monkeypatch):
        import sys
        import types

        cls.opened = []

        def open_ome_zarr(store_path, layout="auto", mode="r", channel_names=None, version="0.4", **kw):
            plate = cls._Plate(store_path, mode, channel_names, version)
            cls.opened.append((str(store_path), layout, mode))
            return plate

        iohub = types.ModuleType("iohub")
        iohub.open_ome_zarr = open_ome_zarr
        ngff = types.ModuleType("iohub.ngff")
        ngff.open_ome_zarr = open_ome_zarr
        models = types.ModuleType("iohub.ngff.models")
        models.TransformationMeta = cls.TransformationMeta
        ngff.models = models
        iohub.ngff = ngff
        for name, mod in (("iohub", iohub), ("iohub.ngff", ngff), ("iohub.ngff.models", models)):
            monkeypatch.setitem(sys.modules, name, mod)


def test_run_store_works_through_the_iohub_surface(tmp_path, cpu_cli, monkeypatch):
    """With ``iohub`` importable (a stand-in with the reference's call surface and NOTHING of this
    package's own array interface), ``open_ome_zarr(prefer_iohub=True)`` delegates and
    ``run_store(io_backend="iohub")`` reads, reconstructs and writes through it."""
    from shrimpy_amd.io.omezarr import Plate, as_volume_array

    _FakeIohub.install(monkeypatch)
    rng = np.random.default_rng(9)
    src = tmp_path / "in.store"
    plate = open_ome_zarr(src, layout="hcs", mode="w", channel_names=["BF", "GFP"], version="0.5")
    assert not isinstance(plate, Plate) and _FakeIohub.opened[-1] == (str(src), "hcs", "w")
    vols = {}
    for key in ("A/1/fov0", "A/2/fov0"):
        pos = plate.create_position(*key.split("/"))
        arr = pos.create_zeros("0", shape=(2, 2, 8, 4, 6), dtype="uint16",
                               transform=[_FakeIohub.TransformationMeta(type="scale", scale=[1, 1, 0.15, 0.11, 0.11])])
        assert not hasattr(arr, "read_volume")
        vols[key] = rng.integers(80, 600, (2, 2, 8, 4, 6)).astype(np.uint16)
        arr[...] = vols[key]
        np.testing.assert_array_equal(as_volume_array(arr).read_volume(1, 0), vols[key][1, 0])
    res = cpu_cli.run_store(src, tmp_path / "out.store", _settings(), zarr_version="0.5",
                            reconstructor_factory=_FakeReconstructor, io_backend="iohub")
    assert res["units"] == 8
    assert (str(tmp_path / "out.store"), "hcs", "w") in _FakeIohub.opened
    out = open_ome_zarr(tmp_path / "out.store", layout="hcs", mode="r")
    for key, pos in out.positions():
        assert pos.channel_names == ["BF", "GFP"]
        assert pos.zattrs["multiscales"][0]["datasets"][0]["coordinateTransformations"][0]["scale"][2:] == [0.15, 0.11, 0.11]
        np.testing.assert_array_equal(pos.data[...], vols[key][:, :, ::2].astype(np.float32) + 0.5)
    # resume goes through the same surface
    res = cpu_cli.run_store(src, tmp_path / "out.store", _settings(), zarr_version="0.5",
                            reconstructor_factory=_FakeReconstructor, io_backend="iohub", resume=True)
    assert res["units_skipped"] == 8


def test_auto_backend_falls_back_to_iohub_only_for_codecs_the_native_reader_lacks(tmp_path, cpu_cli, monkeypatch):
    import click

    src = _make_plate(tmp_path / "raw.zarr", "0.4")
    res = cpu_cli.run_store(src, tmp_path / "o1.zarr", _settings(), reconstructor_factory=_FakeReconstructor)
    assert res["units"] == res["units_total"]                   # native: iohub never needed
    meta = src / "0" / "0" / "000" / "0" / ".zarray"
    m = json.loads(meta.read_text())
    m["compressor"] = {"id": "lzma"}
    meta.write_text(json.dumps(m))
    with pytest.raises(click.ClickException, match="install iohub"):
        cpu_cli.run_store(src, tmp_path / "o2.zarr", _settings(), reconstructor_factory=_FakeReconstructor)
    _FakeIohub.install(monkeypatch)
    with pytest.raises(FileNotFoundError):    # auto now reaches iohub (the stand-in cannot parse a real store)
        cpu_cli.run_store(src, tmp_path / "o3.zarr", _settings(), reconstructor_factory=_FakeReconstructor)
    assert _FakeIohub.opened and _FakeIohub.opened[-1][0] == str(src)


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


@pytest.mark.gpu
def test_cli_deconvolve_with_a_measured_bead_psf_from_a_store_on_gpu(tmp_path):
    """``deconvolve`` with ``psf_path`` naming an OME-Zarr bead volume the size the PSF tools around the reference
    average (15 x 18 x 18, ``scripts/measure_psf.py:187-190, 273-287``): beyond the stencil kernels, so the plan runs the
    iteration in the Fourier domain; store to store against the oracle's direct stencil."""
    from click.testing import CliRunner

    from oracle import cpu_ref as o
    from shrimpy_amd.cli import cli

    rng = np.random.default_rng(21)
    z, y, x = np.meshgrid(np.arange(15) - 7, np.arange(18) - 9, np.arange(18) - 9, indexing="ij")
    beads = np.exp(-0.5 * (((0.9 * z + 0.43 * x) / 2.5) ** 2 + (y / 2.0) ** 2 + ((-0.43 * z + 0.9 * x) / 2.0) ** 2))
    beads = (beads * (1 + 0.02 * rng.standard_normal(beads.shape))).clip(0).astype(np.float32)
    beads /= beads.sum()
    with open_ome_zarr(tmp_path / "psf.zarr", layout="hcs", mode="w", channel_names=["beads"], prefer_iohub=False) as store:
        arr = store.create_position("0", "0", "0").create_zeros("0", shape=(1, 1) + beads.shape, dtype="float32",
                                                                  scale=(1, 1, 0.17, 0.1133, 0.1133))
        arr.write_volume(0, 0, beads)
    src = tmp_path / "deskewed.zarr"
    vols = {}
    with open_ome_zarr(src, layout="hcs", mode="w", channel_names=["LS"], version="0.5", prefer_iohub=False) as plate:
        for key in KEYS[:2]:
            pos = plate.create_position(*key.split("/"))
            arr = pos.create_zeros("0", shape=(1, 1, 12, 30, 44), dtype="float32", scale=(1, 1, 0.17, 0.1133, 0.1133))
            vols[key] = o.bead_scene((12, 30, 44), seed=len(vols) + 3, psf=None, density=4e-3)
            arr.write_volume(0, 0, vols[key])
    (tmp_path / "deconvolve.yml").write_text(yaml.safe_dump(dict(iterations=3, psf_path=str(tmp_path / "psf.zarr"))))
    r = CliRunner().invoke(cli, ["deconvolve", "-i", str(src), "-c", str(tmp_path / "deconvolve.yml"), "-o",
                                 str(tmp_path / "x.zarr")])
    assert r.exit_code == 0, (r.output, r.exception)
    psf = np.pad(beads, ((0, 0), (0, 1), (0, 1)))          # even extents get one trailing zero plane (prepare_psf)
    with open_ome_zarr(tmp_path / "x.zarr", prefer_iohub=False) as out:
        for key, pos in out.positions():
            ref = o.richardson_lucy(vols[key], psf, 3).astype(np.float64)
            got = pos["0"].read_volume(0, 0).astype(np.float64)
            assert np.all(np.abs(got - ref) <= 2e-4 * np.abs(ref) + 1e-4 * np.abs(ref).max())


@pytest.mark.gpu
def test_staged_run_over_a_store_in_the_acquisition_format_on_gpu(tmp_path):
    """deskew + 3 RL iterations store to store with the input as the acquisition writes it -- Zarr v3,
    one shard per (t, c) volume around blosc-zstd chunks (``shrimpy/mantis/mantis_engine.py:474-481``)
    -- through the staged path (reader threads decode the frames into the pinned slots); every
    position against the oracle."""
    from oracle import cpu_ref as o
    from shrimpy_amd import cli as lsr_cli
    from shrimpy_amd.settings import ReconstructSettings

    rng = np.random.default_rng(5)
    src = tmp_path / "acq.ome.zarr"
    vols = {}
    with open_ome_zarr(src, layout="hcs", mode="w", channel_names=["LS"], version="0.5", prefer_iohub=False) as plate:
        for p in range(4):
            arr = plate.create_position("A", str(p + 1), "0").create_zeros(
                "0", shape=(1, 1, 72, 16, 40), dtype="uint16", scale=(1, 1, 0.15, 0.1133, 0.1133),
                chunks=(1, 1, 32, 16, 40), compress="blosc-zstd", shards="volume")
            vols[f"A/{p + 1}/0"] = rng.poisson(300, (72, 16, 40)).astype(np.uint16)
            arr.write_volume(0, 0, vols[f"A/{p + 1}/0"])
    settings = ReconstructSettings.model_validate(dict(
        deskew=dict(pixel_size_um=0.1133, ls_angle_deg=30.0, scan_step_um=0.15, keep_overhang=True, average_n_slices=3),
        deconvolution=dict(iterations=3, gaussian_shape_zyx=[5, 5, 5], gaussian_sigma_zyx=[1.2, 1.0, 1.0])))
    res = lsr_cli.run_store(src, tmp_path / "out.zarr", settings)
    assert res["units_total"] == 4
    psf, _ = o.gaussian_psf((5, 5, 5), (1.2, 1.0, 1.0))
    with open_ome_zarr(tmp_path / "out.zarr", prefer_iohub=False) as plate:
        for key, pos in plate.positions():
            got = pos["0"].read_volume(0, 0).astype(np.float64)
            ref = o.richardson_lucy(o.deskew(vols[key].astype(np.float32), 30.0, 0.755, True, 3), psf, 3).astype(np.float64)
            assert got.shape == ref.shape
            assert np.all(np.abs(got - ref) <= 5e-5 * np.abs(ref) + 2e-5 * np.abs(ref).max())


def test_dash_i_takes_several_position_directories_through_clicks_public_parser(tmp_path, cpu_cli, monkeypatch):
    """``-i a b c`` was parsed by patching click's private parser tables.  It is now a rewrite of the argument
    list in ``Command.parse_args`` (public): several directories after one ``-i``, ``-i`` given twice, the list ended by
    the next option, ``-s`` / ``-t`` of ``register`` alike."""
    from click.testing import CliRunner

    src = _make_plate(tmp_path / "raw.zarr", "0.4")
    seen = {}

    def fake_run_store(input_path, output_path, settings, positions, zarr_version, **kw):
        seen.update(input=input_path, positions=tuple(positions))
        return {"ok": True}

    monkeypatch.setattr(cpu_cli, "run_store", fake_run_store)
    cfg = tmp_path / "dec.yml"
    cfg.write_text(yaml.safe_dump(dict(iterations=2)))
    dirs = [str(src / k) for k in KEYS]
    r = CliRunner().invoke(cpu_cli.cli, ["deconvolve", "-i", *dirs, "-c", str(cfg), "-o", str(tmp_path / "o")])
    assert r.exit_code == 0, r.output
    assert seen["input"] == src and seen["positions"] == tuple(KEYS)
    r = CliRunner().invoke(cpu_cli.cli, ["deconvolve", "-c", str(cfg), "-i", dirs[1], "-i", dirs[0], "-o", str(tmp_path / "o")])
    assert r.exit_code == 0 and set(seen["positions"]) == set(KEYS)
    r = CliRunner().invoke(cpu_cli.cli, ["deconvolve", "-i", str(src), "-c", str(cfg), "-o", str(tmp_path / "o")])
    assert r.exit_code == 0 and seen["input"] == src and seen["positions"] == ()
    reg = tmp_path / "reg.yml"
    reg.write_text(yaml.safe_dump(dict(affine_transform_zyx=np.eye(4).tolist())))
    r = CliRunner().invoke(cpu_cli.cli, ["register", "-s", *dirs, "-t", dirs[0], "-c", str(reg), "-o", str(tmp_path / "o")])
    assert r.exit_code == 0, r.output
    assert seen["input"] == src and seen["positions"] == tuple(KEYS)
