"""``deskew`` / ``register`` / ``deconvolve`` / ``reconstruct`` command line (click), YAML-configured.

The north-star keeps "the existing deskew/register/deconvolve CLI + YAML-config surface" of the
CPU path (biahub's; the reference's own CLI has only ``acquire`` and ``gui``,
``shrimpy/cli/main.py:31-33``).  Idiom follows the reference: a click group with ``-h/--help``
(``shrimpy/cli/main.py:21-33``), strictly validated YAML settings (``shrimpy/config.py:82-162``).

    python -m shrimpy_amd.cli deskew      -i raw.zarr -c deskew.yml      -o deskewed.zarr
    python -m shrimpy_amd.cli register    -i in.zarr  -c register.yml    -o registered.zarr
    python -m shrimpy_amd.cli deconvolve  -i in.zarr  -c deconvolve.yml  -o deconvolved.zarr
    python -m shrimpy_amd.cli reconstruct -i raw.zarr -c recon.yml       -o recon.zarr

Every (position, timepoint, channel) volume is an independent unit.  Launched under
``python -m torch.distributed.run --nproc-per-node N`` each rank takes the units
``rank, rank + N, ...`` on GPU ``LOCAL_RANK`` and writes its own chunks: no data-path collective.
"""

from __future__ import annotations

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
CONTEXT = {"help_option_names": ["-h", "--help"]}


def _common(fn):
    fn = click.option("-i", "--input-position-dirpaths", "input_path", required=True,
                      type=click.Path(exists=True, path_type=Path),
                      help="Input OME-Zarr store (HCS plate or single FOV).")(fn)
    fn = click.option("-c", "--config-filepath", "config", required=True,
                      type=click.Path(exists=True, dir_okay=False, path_type=Path),
                      help="YAML settings file.")(fn)
    fn = click.option("-o", "--output-dirpath", "output_path", required=True,
                      type=click.Path(path_type=Path), help="Output OME-Zarr store (must not exist).")(fn)
    fn = click.option("-p", "--position", "positions", multiple=True,
                      help='Restrict to these position keys ("row/col/fov"); repeatable.')(fn)
    fn = click.option("--zarr-version", type=click.Choice(["0.4", "0.5"]), default="0.4", show_default=True,
                      help="NGFF version of the output store.")(fn)
    return fn


def _distributed():
    """(rank, world, device) from the torchrun environment; initialises the process group for N>1."""
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise click.ClickException("no HIP device: this path runs only on a GPU (MI355X); there is no CPU fallback")
    n_dev = torch.cuda.device_count()
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    torch.cuda.set_device(local % n_dev)
    device = torch.device("cuda", local % n_dev)
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
    return rank, world, device


def run_store(input_path: Path, output_path: Path, settings: ReconstructSettings, positions=(),
              zarr_version: str = "0.4", reconstructor_factory=None, stage_through_pinned: bool = True) -> dict:
    """Apply ``settings`` to every (position, t, c) volume of ``input_path`` -> ``output_path``.

    On a GPU the volumes pass through pinned staging slots and copy streams
    (``staging.VolumeStager``) so that reading, upload, kernels, download and writing overlap.

    ``reconstructor_factory(raw_shape, settings, device)`` defaults to
    :class:`shrimpy_amd.pipeline.VolumeReconstructor` (tests inject a stand-in).
    """
    import torch

    from .io.omezarr import open_ome_zarr
    from .pipeline import Unit, VolumeReconstructor, enumerate_units, run_sharded

    rank, world, device = _distributed()
    factory = reconstructor_factory or VolumeReconstructor
    src = open_ome_zarr(input_path, layout="auto", mode="r")
    src_positions = dict(src.positions())
    keys = [k for k in src_positions if not positions or k in positions]
    missing = [p for p in positions if p not in src_positions]
    if missing:
        raise click.ClickException(f"positions {missing} not found; available: {list(src_positions)}")
    if not keys:
        raise click.ClickException("no positions to process")

    first = src_positions[keys[0]]
    nt, nc, nz, ny, nx = first["0"].shape
    rec = factory((nz, ny, nx), settings, device)
    oz, oy, ox = rec.output_shape
    in_scale = first.scale
    out_scale = list(in_scale)
    if settings.deskew is not None:
        from .geometry import deskew_geometry

        d = settings.deskew
        voxel = deskew_geometry((nz, ny, nx), d.ls_angle_deg, d.px_to_scan_ratio, d.keep_overhang,
                                d.average_n_slices, d.pixel_size_um).voxel_size
        from .geometry import orient_voxel

        # scale metadata as scripts/measure_psf.py:273-276
        out_scale[2:] = [float(v) for v in orient_voxel(voxel, d.orientation)]

    # rank 0 creates the output store (positions are separate arrays on disk); then everybody writes
    if rank == 0:
        dst = open_ome_zarr(output_path, layout="hcs", mode="w", channel_names=first.channel_names,
                            version=zarr_version, prefer_iohub=False)
        for key in keys:
            row, col, fov = key.split("/")
            pos = dst.create_position(row, col, fov)
            pos.create_zeros("0", shape=(nt, nc, oz, oy, ox), dtype="float32", scale=out_scale)
        dst.close()
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
    dst = open_ome_zarr(output_path, layout="hcs", mode="a", prefer_iohub=False)
    dst_positions = dict(dst.positions())

    units = enumerate_units(keys, nt, range(nc))

    def load(u: Unit, out=None):
        return src_positions[u.position]["0"].read_volume(u.t, u.c, out=out)

    def store(u: Unit, vol):
        host = vol if isinstance(vol, np.ndarray) else vol.cpu().numpy()
        dst_positions[u.position]["0"].write_volume(u.t, u.c, host)

    stager = None
    raw_dtype = np.dtype(first["0"].dtype)
    if (stage_through_pinned and torch.device(device).type == "cuda"
            and raw_dtype in (np.dtype("uint16"), np.dtype("float32")) and len(units) > world):
        from .staging import VolumeStager

        try:
            stager = VolumeStager((nz, ny, nx), raw_dtype, (oz, oy, ox), device)
        except (RuntimeError, MemoryError) as exc:   # not enough pinnable host memory for the slots
            logger.warning("staging slots unavailable (%s): volumes are handed over synchronously", exc)
            stager = None
    report = run_sharded(units, load, rec, store, synchronize=torch.cuda.synchronize, stager=stager)
    nvox = len(report.units) * nz * ny * nx
    logger.info("rank %d: %d units, %.3g input voxels/s (job %.2fs)", rank, len(report.units),
                nvox / max(report.seconds, 1e-9), report.max_seconds)
    return {"rank": rank, "world_size": world, "units": len(report.units), "units_total": len(units),
            "seconds": report.seconds, "job_seconds": report.max_seconds, "output_shape": (oz, oy, ox)}


@click.group(context_settings=CONTEXT)
@click.option("-v", "--verbose", is_flag=True, help="DEBUG logging.")
def cli(verbose: bool):
    """MI355X light-sheet reconstruction: deskew, affine registration, Richardson-Lucy."""
    logging.basicConfig(level=logging.DEBUG if verbose else logging.INFO,
                        format="%(asctime)s %(levelname)s %(name)s: %(message)s")


@cli.command()
@_common
def deskew(input_path, config, output_path, positions, zarr_version):
    """Deskew oblique-plane stacks (config: DeskewSettings YAML)."""
    s = ReconstructSettings(deskew=DeskewSettings.from_yaml(config))
    click.echo(run_store(input_path, output_path, s, positions, zarr_version))


@cli.command()
@_common
def register(input_path, config, output_path, positions, zarr_version):
    """Apply an affine registration (config: RegisterSettings YAML with affine_transform_zyx)."""
    s = ReconstructSettings(registration=RegisterSettings.from_yaml(config))
    click.echo(run_store(input_path, output_path, s, positions, zarr_version))


@cli.command()
@_common
def deconvolve(input_path, config, output_path, positions, zarr_version):
    """Richardson-Lucy deconvolution (config: DeconvolveSettings YAML)."""
    s = ReconstructSettings(deconvolution=DeconvolveSettings.from_yaml(config))
    click.echo(run_store(input_path, output_path, s, positions, zarr_version))


@cli.command()
@_common
def reconstruct(input_path, config, output_path, positions, zarr_version):
    """deskew -> register -> deconvolve in one pass (config: ReconstructSettings YAML)."""
    s = ReconstructSettings.from_yaml(config)
    if s.deskew is None and s.registration is None and s.deconvolution is None:
        raise click.ClickException("the config enables no step")
    click.echo(run_store(input_path, output_path, s, positions, zarr_version))


def main():
    cli()


if __name__ == "__main__":
    main()
