"""``deskew`` / ``register`` / ``deconvolve`` / ``reconstruct`` command line (click), YAML-configured.

The north-star keeps "the existing deskew/register/deconvolve CLI + YAML-config surface" of the
CPU path (biahub's; the reference's own CLI has only ``acquire`` and ``gui``,
``shrimpy/cli/main.py:31-33``).  Idiom follows the reference: a click group with ``-h/--help``
(``shrimpy/cli/main.py:21-33``), strictly validated YAML settings (``shrimpy/config.py:82-162``).

    python -m shrimpy_amd.cli deskew      -i raw.zarr -c deskew.yml      -o deskewed.zarr
    python -m shrimpy_amd.cli register    -i in.zarr  -c register.yml    -o registered.zarr
    python -m shrimpy_amd.cli deconvolve  -i in.zarr  -c deconvolve.yml  -o deconvolved.zarr
    python -m shrimpy_amd.cli reconstruct -i raw.zarr -c recon.yml       -o recon.zarr
    python -m shrimpy_amd.cli estimate-registration -s moving.zarr -t target.zarr -o register.yml

Every (position, timepoint, channel) volume is an independent unit.  Launched under
``python -m torch.distributed.run --nproc-per-node N`` each rank takes the units
``rank, rank + N, ...`` on GPU ``LOCAL_RANK`` and writes its own chunks: no data-path collective.
"""

from __future__ import annotations

import functools
import logging
import os

from pathlib import Path

import click
import numpy as np

from .settings import (
    DeconvolveSettings,
    DeskewSettings,
    ReconstructSettings,
    RegisterSettings,
)

logger = logging.getLogger("shrimpy_amd")
# RCCL between the ranks of a node needs dmabuf IPC on hosts whose driver supports nothing else (otherwise
# hipIpcGetMemHandle: invalid argument).  The HSA runtime reads this when the first HIP call initialises it, i.e. before
# anything below can run -- so it is set when the CLI module is imported, and only if the caller has not decided.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
CONTEXT = {"help_option_names": ["-h", "--help"]}


class _EatAllCommand(click.Command):
    """A command whose ``eat_all`` options take every following argument up to the next option -- what biahub's ``-i``
    does ([RECALLED] ``OptionEatAll``), so that a shell glob works: ``-i plate.zarr/*/*/*`` expands to one path per
    position directory.  Done with click's public surface only (the earlier version patched the parser's
    private tables): ``parse_args`` rewrites ``-i a b c`` into ``-i a -i b -i c`` before click sees it, and the options
    are ordinary ``multiple=True`` ones.  An argument that starts with ``-`` ends the list (a directory named ``-x`` has
    to be written ``./-x``)."""

    eat_all: tuple = ()

    def parse_args(self, ctx, args):
        out, i = [], 0
        args = list(args)
        while i < len(args):
            a = args[i]
            out.append(a)
            i += 1
            if a == "--":                       # everything behind it is positional
                out.extend(args[i:])
                break
            if a in self.eat_all:
                first = True
                while i < len(args) and not (args[i].startswith("-") and len(args[i]) > 1):
                    if not first:
                        out.append(a)
                    out.append(args[i])
                    first = False
                    i += 1
        return super().parse_args(ctx, out)


def _eat_all_command(*names):
    """``cls=`` for ``@cli.command``: an :class:`_EatAllCommand` whose options ``names`` eat every following argument."""
    return type("_EatAllCommand_" + "_".join(n.strip("-").replace("-", "_") for n in names), (_EatAllCommand,), {"eat_all": tuple(names)})


def resolve_inputs(paths) -> tuple[Path, tuple[str, ...]]:
    """``(store root, position keys)`` of what ``-i`` was given: ONE store (an HCS plate or a single FOV: every
    position, keys ``()``) or N position directories ``<plate>/<row>/<col>/<fov>`` of one plate (the expansion of
    ``plate.zarr/*/*/*``), grouped by their common plate root."""
    from .io.omezarr import _detect

    paths = [Path(p) for p in ((paths,) if isinstance(paths, (str, os.PathLike)) else paths)]
    if not paths:
        raise click.ClickException("-i needs at least one path")
    for p in paths:
        if not p.exists():
            raise click.ClickException(f"-i: {p} does not exist")
    if len(paths) == 1:
        p = paths[0]
        try:
            _, layout = _detect(p)
        except (FileNotFoundError, ValueError) as exc:
            raise click.ClickException(f"-i: {exc}") from exc
        if layout == "fov" and len(p.resolve().parents) >= 3:
            try:   # a position directory of a plate: process that position of the plate (its key is the plate's)
                root = p.resolve().parents[2]
                if _detect(root)[1] == "hcs":
                    return root, ("/".join(p.resolve().parts[-3:]),)
            except (FileNotFoundError, ValueError):
                pass
        return p, ()
    roots, keys = set(), []
    for p in paths:
        r = p.resolve()
        if len(r.parents) < 3:
            raise click.ClickException(f"-i: {p} is not a <plate>/<row>/<col>/<fov> position directory")
        roots.add(r.parents[2])
        keys.append("/".join(r.parts[-3:]))
    if len(roots) != 1:
        raise click.ClickException(f"-i: the position directories belong to {len(roots)} different plates "
                                   f"({sorted(str(r) for r in roots)}); run them one plate at a time")
    root = roots.pop()
    try:
        if _detect(root)[1] != "hcs":
            raise ValueError(f"{root} is not an HCS plate")
    except (FileNotFoundError, ValueError) as exc:
        raise click.ClickException(f"-i: several paths must be position directories of one plate: {exc}") from exc
    if len(set(keys)) != len(keys):
        raise click.ClickException("-i: a position directory is listed twice")
    return root, tuple(keys)


def _common(fn, input_required: bool = True):
    fn = click.option("-i", "--input-position-dirpaths", "input_path", required=input_required, multiple=True,
                      type=click.UNPROCESSED,
                      help="Input OME-Zarr store (HCS plate or single FOV), or the position directories of one plate "
                           "(a glob such as plate.zarr/*/*/*).")(fn)
    fn = click.option("-c", "--config-filepath", "config", required=True,
                      type=click.Path(exists=True, dir_okay=False, path_type=Path),
                      help="YAML settings file.")(fn)
    fn = click.option("-o", "--output-dirpath", "output_path", required=True,
                      type=click.Path(path_type=Path),
                      help="Output OME-Zarr store (must not exist, unless --resume).")(fn)
    fn = click.option("-p", "--position", "positions", multiple=True,
                      help='Restrict to these position keys ("row/col/fov"); repeatable.')(fn)
    fn = click.option("--zarr-version", type=click.Choice(["0.4", "0.5"]), default="0.5", show_default=True,
                      help="NGFF version of the output store (0.5 = Zarr v3, what the reference writes: "
                           "shrimpy/dynatrack/tracking.py:1337-1343).")(fn)
    fn = click.option("--resume", is_flag=True,
                      help="Continue an interrupted run: units already completed in the output store "
                           "(same input and settings) are skipped, unfinished ones are rewritten.")(fn)
    fn = click.option("--io", "io_backend", type=click.Choice(["auto", "native", "iohub"]), default="auto",
                      show_default=True,
                      help="Store access: this package's reader/writer, iohub, or native with iohub as the "
                           "fallback for codecs it does not implement.")(fn)
    fn = click.option("--compression", type=click.Choice(["none", "gzip", "zstd", "blosc-zstd"]),
                      default="blosc-zstd", show_default=True,
                      help="Chunk compression of the output (native writer); blosc-zstd is what the acquisition engine "
                           "writes (shrimpy/mantis/mantis_engine.py:474-481), none is the fastest.")(fn)
    fn = click.option("--on-error", "on_error", type=click.Choice(["raise", "skip"]), default="raise", show_default=True,
                      help="skip: a unit whose chunks do not read, whose kernels fail or whose result cannot be written is "
                           "left out and recorded under <output>/.lsr_failed/ (every other unit is written; the command "
                           "exits with status 3 and --resume retries only the failed units) -- the reference likewise "
                           "turns a failed stack into an error record and carries on (shrimpy/dynatrack/worker.py:262-271).")(fn)
    return fn


def _finish(result: dict) -> None:
    """Print a command's result; exit with status 3 when units were skipped (``--on-error skip``)."""
    click.echo(result)
    failed = result.get("failed") or []
    if failed:
        for f in failed:
            click.echo(f"FAILED {f['position']} t={f['t']} c={f['c']} [{f['stage']}]: {f['error']}", err=True)
        raise SystemExit(3)


def _inputs(input_path, positions):
    """The store and position selection of a command: ``-i`` (one store or N position directories) and ``-p``."""
    root, keys = resolve_inputs(input_path)
    if keys and positions:
        missing = [p for p in positions if p not in keys]
        if missing:
            raise click.ClickException(f"-p {missing}: not among the position directories given to -i")
        keys = tuple(k for k in keys if k in positions)
    return root, tuple(keys) or tuple(positions)


def _distributed():
    """(rank, world, device, created): ranks from the torchrun environment; the process group is
    initialised here for N > 1 (``created`` tells the caller to destroy it again)."""
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        # the reference's own rule (shrimpy/preprocessing.py:78-82): no GPU visible -> cpu.  The volumes then
        # run the native host twins (shrimpy_amd/host.py); ranks, if any, meet over gloo.
        created = False
        if world > 1:
            import torch.distributed as dist

            if not dist.is_initialized():
                dist.init_process_group(os.environ.get("LSR_DIST_BACKEND") or "gloo")
                created = True
        return rank, world, torch.device("cpu"), created
    n_dev = torch.cuda.device_count()
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    torch.cuda.set_device(local % n_dev)
    device = torch.device("cuda", local % n_dev)
    created = False
    if world > 1:
        import torch.distributed as dist

        if not dist.is_initialized():
            # one rank per GPU: RCCL.  More ranks than GPUs on the node (ranks share a card, e.g. a
            # rehearsal on a one-GPU box): RCCL refuses duplicate devices, gloo carries the barrier
            # and the timing reduction instead -- the data path has no collective either way.
            backend = os.environ.get("LSR_DIST_BACKEND") or ("nccl" if local_world <= n_dev else "gloo")
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=device)
            else:
                dist.init_process_group(backend)
            created = True
    return rank, world, device, created


def _agree(error: str | None) -> None:
    """Every rank learns whether rank 0's set-up step failed, and all of them stop together --
    a bare barrier would leave the others waiting for the collective timeout."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        box = [error]
        dist.broadcast_object_list(box, src=0)
        error = box[0]
    if error:
        raise click.ClickException(error)


def _open_source(path: Path, io_backend: str):
    """``(store, {key: position})`` of the input, through the native reader or iohub.

    ``auto`` = native first (it reads chunks straight into pinned staging slots); when an array uses
    a codec it does not implement and iohub is importable, the whole store is reopened with iohub."""
    from .io.omezarr import UnsupportedCodec, open_ome_zarr

    def positions_of(store):
        return dict(store.positions()) if hasattr(store, "positions") else {"0/0/0": store}

    if io_backend in ("auto", "native"):
        store = open_ome_zarr(path, layout="auto", mode="r", prefer_iohub=False)
        pos = positions_of(store)
        try:
            for p in pos.values():
                p["0"]          # parses the array metadata: raises on an unknown codec
            return store, pos
        except UnsupportedCodec as exc:
            if io_backend == "native":
                raise click.ClickException(str(exc)) from exc
            try:
                import iohub  # noqa: F401
            except ImportError:
                raise click.ClickException(f"{exc} (iohub is not installed either)") from exc
    try:
        from iohub import open_ome_zarr as iohub_open
    except ImportError as exc:
        raise click.ClickException("--io iohub: iohub is not installed") from exc
    store = iohub_open(str(path), layout="auto", mode="r")
    return store, positions_of(store)


def _channel_plan(settings: ReconstructSettings, names: list[str], nc: int) -> list[bool]:
    """Per channel: does the registration step apply?  ``source_channel_names`` empty = every channel
    (the single-modality case); otherwise only the named ones are warped and every other channel --
    the target included -- goes through the remaining steps unwarped."""
    reg = settings.registration
    if reg is None:
        return [False] * nc
    if not reg.source_channel_names:
        if reg.target_channel_name is not None and nc > 1:
            raise click.ClickException(
                "registration names a target channel but no source_channel_names: list the channels to warp")
        return [True] * nc
    known = list(names) or [str(i) for i in range(nc)]
    wanted = list(reg.source_channel_names) + ([reg.target_channel_name] if reg.target_channel_name else [])
    missing = [n for n in wanted if n not in known]
    if missing:
        raise click.ClickException(f"channels {missing} not in the store (it has {known})")
    if reg.target_channel_name in reg.source_channel_names:
        raise click.ClickException(f"target channel {reg.target_channel_name!r} is also listed as a source")
    return [known[c] in reg.source_channel_names for c in range(nc)]


def _fingerprint(input_path: Path, settings: ReconstructSettings, shape, dtype, keys) -> str:
    import hashlib
    import json

    doc = {"input": str(Path(input_path).resolve()), "settings": settings.model_dump(mode="json"),
           "shape": list(shape), "dtype": str(dtype), "positions": list(keys)}
    return hashlib.sha256(json.dumps(doc, sort_keys=True).encode()).hexdigest()


class _DoneLedger:
    """Per-unit completion records of an output store: one small file per (position, t, c) under
    ``<store>/.lsr_done/``, written only after the unit's chunks are.

    The reference never overwrites an acquisition and appends its CSV row by row
    (``shrimpy/mantis/mantis_engine.py:458, 480``, ``shrimpy/dynatrack/tracking.py:887-914``); the
    unit of resumption here is likewise the per-(p, t, c) volume.  A record carries
    the fingerprint of (input, settings, geometry): a store written with other settings is never
    silently continued."""

    def __init__(self, store_path: Path, fingerprint: str):
        self.dir = Path(store_path) / ".lsr_done"
        self.fingerprint = fingerprint

    def _file(self, u) -> Path:
        return self.dir / u.position.replace("/", "__") / f"t{u.t}_c{u.c}"

    def begin(self) -> None:
        self.dir.mkdir(parents=True, exist_ok=True)
        tag = self.dir / "fingerprint"
        if tag.exists():
            if tag.read_text().strip() != self.fingerprint:
                raise click.ClickException(
                    f"{self.dir.parent} was written from a different input or with different settings; "
                    "--resume only continues the same run")
        else:
            tag.write_text(self.fingerprint + "\n")

    def is_done(self, u) -> bool:
        f = self._file(u)
        try:
            return f.read_text().strip() == self.fingerprint
        except FileNotFoundError:
            return False

    def mark(self, u) -> None:
        f = self._file(u)
        f.parent.mkdir(parents=True, exist_ok=True)
        tmp = f.with_name(f.name + ".tmp")
        tmp.write_text(self.fingerprint + "\n")
        os.replace(tmp, f)


class _FailLedger:
    """Units given up under ``--on-error skip``: one JSON file per (position, t, c) under ``<store>/.lsr_failed/`` with
    the stage and the message; removed again when the unit is written by a later (``--resume``) run."""

    def __init__(self, store_path: Path):
        self.dir = Path(store_path) / ".lsr_failed"

    def _file(self, u) -> Path:
        return self.dir / u.position.replace("/", "__") / f"t{u.t}_c{u.c}.json"

    def mark(self, u, stage: str, error: str) -> None:
        import json

        f = self._file(u)
        f.parent.mkdir(parents=True, exist_ok=True)
        f.write_text(json.dumps({"position": u.position, "t": u.t, "c": u.c, "stage": stage, "error": error}) + "\n")

    def clear(self, u) -> None:
        try:
            self._file(u).unlink()
        except FileNotFoundError:
            pass

    def listed(self) -> list[dict]:
        import json

        return [json.loads(f.read_text()) for f in sorted(self.dir.glob("*/*.json"))] if self.dir.exists() else []


def run_store(input_path: Path, output_path: Path, settings: ReconstructSettings, positions=(),
              zarr_version: str = "0.4", reconstructor_factory=None, stage_through_pinned: bool = True,
              resume: bool = False, io_backend: str = "auto", compression: str | None = None,
              device_codec: bool | None = None, on_error: str = "raise") -> dict:
    """Apply ``settings`` to every (position, t, c) volume of ``input_path`` -> ``output_path``.

    ``device_codec`` (default: on, ``LSR_DEVICE_CODEC=0`` turns it off): with a blosc-zstd output on a GPU the chunk
    frames are written by the device (``io/device_codec.py``) and the host stores them as they are; an input stored as
    blosc-zstd frames crosses PCIe compressed and is decoded by the device.  The decoder gives every blosc block to one
    lane, so a volume takes ~6 ms whatever its size: left to itself (``None``) the run decodes on the device only volumes of
    at least ``LSR_DEVICE_DECODE_MIN_BLOCKS`` blocks (default 8192 = 256 MB of the acquisition's 32 KB blocks; below that
    sixteen host threads are faster); ``True`` decodes on the device whatever the size.
    ``on_error="skip"``: see ``pipeline.run_sharded``; skipped units are listed in the result's ``failed`` (all ranks')
    and recorded under ``<output>/.lsr_failed/``.

    On a GPU the volumes pass through pinned staging slots and copy streams
    (``staging.VolumeStager``) so that reading, upload, kernels, download and writing overlap.

    ``reconstructor_factory(raw_shape, settings, device)`` defaults to
    :class:`shrimpy_amd.pipeline.VolumeReconstructor` (tests inject a stand-in).
    ``resume``: the output may exist; units recorded as complete (``_DoneLedger``) are skipped.
    """
    rank, world, device, created = _distributed()
    try:
        min_blocks = 0
        if device_codec is None:
            device_codec = os.environ.get("LSR_DEVICE_CODEC", "1") != "0"
            min_blocks = int(os.environ.get("LSR_DEVICE_DECODE_MIN_BLOCKS", "8192") or 0)
        return _run_store(input_path, output_path, settings, positions, zarr_version, reconstructor_factory,
                          stage_through_pinned, resume, io_backend, compression, rank, world, device, device_codec,
                          on_error, min_blocks)
    finally:
        if created:
            import torch.distributed as dist

            dist.destroy_process_group()


def _run_store(input_path, output_path, settings, positions, zarr_version, reconstructor_factory,
               stage_through_pinned, resume, io_backend, compression, rank, world, device, device_codec=True,
               on_error="raise", min_decode_blocks=0) -> dict:
    import torch

    from .io.omezarr import as_volume_array, create_level, open_ome_zarr, position_scale
    from .pipeline import Unit, VolumeReconstructor, enumerate_units, run_sharded

    factory = reconstructor_factory or VolumeReconstructor
    src, src_positions = _open_source(input_path, io_backend)
    use_iohub_out = io_backend == "iohub"
    keys = [k for k in src_positions if not positions or k in positions]
    missing = [p for p in positions if p not in src_positions]
    if missing:
        raise click.ClickException(f"positions {missing} not found; available: {list(src_positions)}")
    if not keys:
        raise click.ClickException("no positions to process")

    # every selected position must have the first one's geometry: checked before anything is written
    arrays = {k: as_volume_array(src_positions[k]["0"]) for k in keys}
    first = src_positions[keys[0]]
    shape5, raw_dtype = tuple(arrays[keys[0]].shape), np.dtype(arrays[keys[0]].dtype)
    if len(shape5) != 5:
        raise click.ClickException(f"position {keys[0]}: expected 5-D TCZYX data, got shape {shape5}")
    odd = {k: (tuple(a.shape), str(a.dtype)) for k, a in arrays.items()
           if tuple(a.shape) != shape5 or np.dtype(a.dtype) != raw_dtype}
    if odd:
        raise click.ClickException(f"positions differ from {keys[0]} {shape5} {raw_dtype}: {odd}; "
                                   "run them separately (-p)")
    nt, nc, nz, ny, nx = shape5
    channel_names = list(first.channel_names)
    warp = _channel_plan(settings, channel_names, nc)

    rec = factory((nz, ny, nx), settings, device)
    rec_unwarped = rec
    if settings.registration is not None and not all(warp):
        rec_unwarped = factory((nz, ny, nx), settings.model_copy(update={"registration": None}), device)
        if tuple(rec_unwarped.output_shape) != tuple(rec.output_shape):
            hint = ("keep_overhang grows the registered channels to the union box, which an unwarped channel cannot share: "
                    "warp every channel (leave source_channel_names empty) or set keep_overhang: false"
                    if settings.registration.keep_overhang else "drop output_shape_zyx or warp every channel")
            raise click.ClickException(
                f"registered channels come out as {tuple(rec.output_shape)} but the unwarped ones as "
                f"{tuple(rec_unwarped.output_shape)}: {hint}")
    oz, oy, ox = rec.output_shape
    out_scale = list(position_scale(first))
    if settings.deskew is not None:
        from .geometry import deskew_geometry, orient_voxel

        d = settings.deskew
        voxel = deskew_geometry((nz, ny, nx), d.ls_angle_deg, d.px_to_scan_ratio, d.keep_overhang,
                                d.average_n_slices, d.pixel_size_um).voxel_size
        # scale metadata as scripts/measure_psf.py:273-276
        out_scale[2:] = [float(v) for v in orient_voxel(voxel, d.orientation)]

    # where output index 0 sits in the target's physical frame: non-zero only when keep_overhang grew the grid below zero
    origin = tuple(getattr(rec, "register_origin", (0, 0, 0)))
    out_translation = [0.0, 0.0] + [float(o) * float(sc) for o, sc in zip(origin, out_scale[2:])] if any(origin) else None
    ledger = _DoneLedger(output_path, _fingerprint(input_path, settings, shape5, raw_dtype, keys))
    fail_ledger = _FailLedger(output_path)
    out_shape5 = (nt, nc, oz, oy, ox)

    def open_output(mode):
        if use_iohub_out:
            from iohub import open_ome_zarr as iohub_open

            kw = dict(layout="hcs", mode=mode)
            if mode == "w":
                kw.update(channel_names=channel_names, version=zarr_version)
            return iohub_open(str(output_path), **kw)
        return open_ome_zarr(output_path, layout="hcs", mode=mode, channel_names=channel_names,
                             version=zarr_version, prefer_iohub=False)

    # rank 0 creates (or, resuming, checks and completes) the output store; positions are separate
    # arrays on disk, so afterwards every rank writes its own units with no coordination
    error = None
    if rank == 0:
        try:
            exists = Path(output_path).exists() and any(Path(output_path).iterdir())
            if exists and not resume:
                raise FileExistsError(f"{output_path} exists (never overwritten, like the reference); "
                                      "use --resume to continue an interrupted run")
            dst = open_output("a" if exists else "w")
            have = dict(dst.positions()) if exists else {}
            for key in keys:
                if key in have:
                    got = tuple(as_volume_array(have[key]["0"]).shape)
                    if got != out_shape5:
                        raise ValueError(f"{output_path}: position {key} has shape {got}, this run writes {out_shape5}")
                    continue
                row, col, fov = key.split("/")
                pos = dst.create_position(row, col, fov)
                extra = {} if use_iohub_out or compression in (None, "none") else {"compress": compression}
                create_level(pos, out_shape5, "float32", out_scale, translation=out_translation, **extra)
            dst.close()
            ledger.begin()
        except click.ClickException as exc:
            error = exc.message
        except Exception as exc:  # noqa: BLE001 -- reported on every rank, see _agree
            error = f"{type(exc).__name__}: {exc}"
    _agree(error)
    dst = open_output("a")
    dst_arrays = {k: as_volume_array(p["0"]) for k, p in dict(dst.positions()).items() if k in keys}

    units = enumerate_units(keys, nt, range(nc))
    todo = [u for u in units if not (resume and ledger.is_done(u))]
    skipped = len(units) - len(todo)
    if skipped:
        logger.info("resume: %d of %d units already complete", skipped, len(units))

    frames_in = [False]      # set below: the stager decodes the store's chunk frames on the device

    def load(u: Unit, out=None):
        if frames_in[0] and out is not None:
            return arrays[u.position].read_volume_frames(u.t, u.c, out=out)   # file reads only
        return arrays[u.position].read_volume(u.t, u.c, out=out)

    def store(u: Unit, vol):
        from .staging import EncodedVolume

        if isinstance(vol, EncodedVolume):      # chunk frames written on the device: stored as they are
            dst_arrays[u.position].write_encoded_volume(u.t, u.c, vol.frames)
        else:
            host = vol if isinstance(vol, np.ndarray) else vol.cpu().numpy()
            dst_arrays[u.position].write_volume(u.t, u.c, host)
        ledger.mark(u)
        fail_ledger.clear(u)

    def process(data, unit: Unit):
        return (rec if warp[unit.c] else rec_unwarped)(data)

    stager = None
    if (stage_through_pinned and torch.device(device).type == "cuda"
            and raw_dtype in (np.dtype("uint16"), np.dtype("float32")) and len(todo) > world):
        from .staging import VolumeStager

        # a blosc-zstd output whose chunks are whole planes: the device writes the frames (None: host encoding)
        frame_bytes = None
        if device_codec and not use_iohub_out:
            sizes = {getattr(a, "encoded_frame_bytes", lambda: None)() for a in dst_arrays.values()}
            frame_bytes = sizes.pop() if len(sizes) == 1 else None
        # ... and an input stored as blosc-zstd chunk frames of whole planes is decoded on the device
        layout = None
        if device_codec and todo:
            layout = getattr(arrays[todo[0].position], "compressed_layout", lambda *a: None)(todo[0].t, todo[0].c)
            if layout is not None:
                blocks = layout["n_frames"] * -(-layout["nbytes"] // layout["blocksize"])
                if blocks < min_decode_blocks:
                    logger.info("input chunks decoded on the host: %d blosc blocks per volume, the device decoder pays from %d",
                                blocks, min_decode_blocks)
                    layout = None
        from ._lib import LsrUnsupported

        attempts = [(frame_bytes, layout)] + ([(None, layout)] if frame_bytes and layout else []) \
            + ([(frame_bytes, None)] if frame_bytes and layout else []) + ([(None, None)] if frame_bytes or layout else [])
        # slots each way: with the chunks decoded on the device a unit passes through read -> upload -> decode before the
        # kernels can start, and three slots let the reader work two units ahead of them (LSR_STAGE_DEPTH overrides)
        depth = int(os.environ.get("LSR_STAGE_DEPTH", "0") or 0)
        for fb, lay in attempts:
            try:
                stager = VolumeStager((nz, ny, nx), raw_dtype, (oz, oy, ox), device, encode_frame_bytes=fb, decode_layout=lay,
                                      depth=depth if depth >= 2 else (3 if lay is not None else 2))
                frames_in[0] = lay is not None
                break
            except LsrUnsupported as exc:                # a chunk / block size outside the device codecs' range
                logger.info("device codec not used for %s (%s)", "the output" if fb else "the input", exc)
            except (RuntimeError, MemoryError) as exc:   # not enough pinnable host memory for the slots
                logger.warning("staging slots unavailable (%s): volumes are handed over synchronously", exc)
                stager = None
                break
    try:
        on_gpu = torch.device(device).type == "cuda"
        report = run_sharded(todo, load, process, store, synchronize=torch.cuda.synchronize if on_gpu else None,
                             stager=stager, process_takes_unit=True, on_error=on_error)
    finally:
        if stager is not None:
            try:                               # (LSR_STAGE_EVENTS=1 only; never in the way of the run's own error)
                times = stager.gpu_times() if hasattr(stager, "gpu_times") else None
                if times:
                    logger.warning("GPU time per unit by HIP events: %s", times)
            except Exception as exc:  # noqa: BLE001
                logger.debug("no GPU clocks: %s", exc)
            stager.close()
        for s in (src, dst):
            close = getattr(s, "close", None)
            if close:
                close()
    failed = [{"position": u.position, "t": u.t, "c": u.c, "stage": stage, "error": msg} for u, stage, msg in report.failures]
    for u, stage, msg in report.failures:
        fail_ledger.mark(u, stage, msg)
    if world > 1:                       # every rank returns the whole job's list
        import torch.distributed as dist

        box: list = [None] * world
        dist.all_gather_object(box, failed)
        failed = [f for part in box for f in part]
    nvox = (len(report.units) - len(report.failures)) * nz * ny * nx
    logger.info("rank %d: %d units, %.3g input voxels/s (job %.2fs)", rank, len(report.units),
                nvox / max(report.seconds, 1e-9), report.max_seconds)
    return {"rank": rank, "world_size": world, "units": len(report.units), "units_total": len(units),
            "units_skipped": skipped, "failed": failed, "seconds": report.seconds, "job_seconds": report.max_seconds,
            "stage_seconds": {k: round(v, 4) for k, v in report.stage_seconds.items()},
            # median interval between consecutive units on this rank once the pipeline is full (None: fewer than 4 units)
            "steady_s_per_unit": (round(float(np.median(report.unit_intervals[1:])), 4)
                                  if len(report.unit_intervals) >= 3 else None),
            "device_codec": {"encode": bool(stager is not None and getattr(stager, "encode_frame_bytes", None)),
                             "decode": bool(frames_in[0])},
            "output_shape": (oz, oy, ox)}


@click.group(context_settings=CONTEXT)
@click.option("-v", "--verbose", is_flag=True, help="DEBUG logging.")
def cli(verbose: bool):
    """MI355X light-sheet reconstruction: deskew, affine registration, Richardson-Lucy."""
    logging.basicConfig(level=logging.DEBUG if verbose else logging.INFO,
                        format="%(asctime)s %(levelname)s %(name)s: %(message)s")


@cli.command(cls=_eat_all_command("-i", "--input-position-dirpaths"))
@_common
def deskew(input_path, config, output_path, positions, zarr_version, resume, io_backend, compression, on_error):
    """Deskew oblique-plane stacks (config: DeskewSettings YAML)."""
    input_path, positions = _inputs(input_path, positions)
    s = ReconstructSettings(deskew=DeskewSettings.from_yaml(config))
    _finish(run_store(input_path, output_path, s, positions, zarr_version, resume=resume,
                      io_backend=io_backend, compression=compression, on_error=on_error))


def _target_shape_zyx(target_path) -> tuple[int, int, int]:
    """(Z, Y, X) of the first position of the store (or position directories) ``-t`` names."""
    from .io.omezarr import as_volume_array

    root, keys = resolve_inputs(target_path)
    store, positions_of = _open_source(root, "auto")
    try:
        for key, pos in positions_of.items():
            if not keys or key in keys:
                return tuple(int(v) for v in as_volume_array(pos["0"]).shape[-3:])
    finally:
        if hasattr(store, "close"):
            store.close()
    raise click.ClickException(f"-t: no position found in {root}")


@cli.command(cls=_eat_all_command("-i", "--input-position-dirpaths", "-s", "--source-position-dirpaths", "-t", "--target-position-dirpaths"))
@click.option("-s", "--source-position-dirpaths", "source_path", multiple=True, type=click.UNPROCESSED,
              help="The MOVING store / position directories ([RECALLED] biahub's spelling; the same as -i).")
@click.option("-t", "--target-position-dirpaths", "target_path", multiple=True, type=click.UNPROCESSED,
              help="The TARGET store / position directories: its (Z, Y, X) is the output shape when the config gives "
                   "no output_shape_zyx.")
@functools.partial(_common, input_required=False)
def register(input_path, source_path, target_path, config, output_path, positions, zarr_version, resume, io_backend,
             compression, on_error):
    """Apply an affine registration (config: RegisterSettings YAML with affine_transform_zyx)."""
    input_path, source_path, target_path = (input_path or None), (source_path or None), (target_path or None)
    if (input_path is None) == (source_path is None):
        raise click.ClickException("name the moving store once: -i or -s")
    input_path, positions = _inputs(input_path if input_path is not None else source_path, positions)
    reg = RegisterSettings.from_yaml(config)
    if target_path is not None and reg.output_shape_zyx is None:
        reg = reg.model_copy(update={"output_shape_zyx": _target_shape_zyx(target_path)})
    s = ReconstructSettings(registration=reg)
    _finish(run_store(input_path, output_path, s, positions, zarr_version, resume=resume,
                      io_backend=io_backend, compression=compression, on_error=on_error))


@cli.command(cls=_eat_all_command("-i", "--input-position-dirpaths"))
@click.option("--psf-dirpath", "psf_dirpath", default=None, type=click.Path(exists=True, path_type=Path),
              help="A measured PSF: an OME-Zarr bead volume (as scripts/measure_psf.py:273-287 writes them) or a .npy "
                   "ZYX array; overrides psf_path of the config ([RECALLED] biahub's -p, which is the position filter here).")
@_common
def deconvolve(input_path, psf_dirpath, config, output_path, positions, zarr_version, resume, io_backend, compression,
               on_error):
    """Richardson-Lucy deconvolution (config: DeconvolveSettings YAML)."""
    input_path, positions = _inputs(input_path, positions)
    dec = DeconvolveSettings.from_yaml(config)
    if psf_dirpath is not None:
        dec = dec.model_copy(update={"psf_path": str(psf_dirpath)})
    s = ReconstructSettings(deconvolution=dec)
    _finish(run_store(input_path, output_path, s, positions, zarr_version, resume=resume,
                      io_backend=io_backend, compression=compression, on_error=on_error))


@cli.command(cls=_eat_all_command("-i", "--input-position-dirpaths"))
@_common
def reconstruct(input_path, config, output_path, positions, zarr_version, resume, io_backend, compression, on_error):
    """deskew -> register -> deconvolve in one pass (config: ReconstructSettings YAML)."""
    input_path, positions = _inputs(input_path, positions)
    s = ReconstructSettings.from_yaml(config)
    if s.deskew is None and s.registration is None and s.deconvolution is None:
        raise click.ClickException("the config enables no step")
    _finish(run_store(input_path, output_path, s, positions, zarr_version, resume=resume,
                      io_backend=io_backend, compression=compression, on_error=on_error))


@cli.command("estimate-registration")
@click.option("-s", "--source-position-dirpaths", "source_path", required=True,
              type=click.Path(exists=True, path_type=Path), help="Store holding the MOVING volume.")
@click.option("-t", "--target-position-dirpaths", "target_path", required=True,
              type=click.Path(exists=True, path_type=Path), help="Store holding the TARGET (fixed) volume.")
@click.option("-o", "--output-filepath", "output_path", required=True, type=click.Path(dir_okay=False, path_type=Path),
              help="RegisterSettings YAML to write (affine_transform_zyx, source_channel_names).")
@click.option("--source-channel", default=None, help="Channel name in the source store (default: the first).")
@click.option("--target-channel", default=None, help="Channel name in the target store (default: the first).")
@click.option("-p", "--position", default=None, help='Position key ("row/col/fov") in both stores (default: the first).')
@click.option("--timepoint", type=int, default=0, show_default=True)
@click.option("--model", type=click.Choice(["affine", "translation"]), default="affine", show_default=True)
@click.option("--no-intensity", is_flag=True, help="Do not fit the gain / offset between the two volumes.")
@click.option("--io", "io_backend", type=click.Choice(["auto", "native", "iohub"]), default="auto", show_default=True)
def estimate_registration(source_path, target_path, output_path, source_channel, target_channel, position, timepoint,
                          model, no_intensity, io_backend):
    """Estimate the affine that maps TARGET indices to SOURCE coordinates (what `register` applies)."""
    click.echo(run_estimate(source_path, target_path, output_path, source_channel, target_channel, position, timepoint,
                            model, not no_intensity, io_backend))


def run_estimate(source_path, target_path, output_path, source_channel=None, target_channel=None, position=None,
                 timepoint: int = 0, model: str = "affine", intensity: bool = True, io_backend: str = "auto",
                 estimator=None) -> dict:
    """Read one volume from each store, estimate, write the ``RegisterSettings`` YAML.
    ``estimator(moving, target, model=, intensity=)`` defaults to :func:`estimate.estimate_affine_zyx`."""
    import torch
    import yaml

    from .io.omezarr import as_volume_array

    _, _, device, created = _distributed()
    try:
        def read(path, channel):
            _, positions = _open_source(path, io_backend)
            key = position or next(iter(positions))
            if key not in positions:
                raise click.ClickException(f"position {key!r} not in {path} (it has {list(positions)})")
            pos = positions[key]
            names = list(pos.channel_names)
            if channel is None:
                c = 0
            elif channel in names:
                c = names.index(channel)
            else:
                raise click.ClickException(f"channel {channel!r} not in {path} (it has {names})")
            arr = as_volume_array(pos["0"])
            if not 0 <= timepoint < arr.shape[0]:
                raise click.ClickException(f"timepoint {timepoint} out of range for {path} (T = {arr.shape[0]})")
            vol = arr.read_volume(timepoint, c)
            return torch.as_tensor(np.ascontiguousarray(vol, dtype=np.float32), device=device), (names[c] if names else str(c))

        moving, src_name = read(source_path, source_channel)
        target, tgt_name = read(target_path, target_channel)
        if estimator is None:
            from .estimate import estimate_affine_zyx as estimator
        est = estimator(moving, target, model=model, intensity=intensity)
        doc = est.to_settings_dict(source_channel_names=[src_name], output_shape_zyx=[int(n) for n in target.shape])
        if Path(source_path).resolve() == Path(target_path).resolve() and tgt_name != src_name:
            doc["target_channel_name"] = tgt_name
        RegisterSettings(**doc)   # what we write must load
        Path(output_path).parent.mkdir(parents=True, exist_ok=True)
        with open(output_path, "w") as f:
            yaml.safe_dump(doc, f, sort_keys=False)
        return {"output": str(output_path), "rms": est.rms, "gain": est.gain, "offset": est.offset,
                "iterations": est.iterations, "converged": est.converged,
                "affine_transform_zyx": doc["affine_transform_zyx"]}
    finally:
        if created:
            import torch.distributed as dist

            dist.destroy_process_group()


def main():
    cli()


if __name__ == "__main__":
    main()
