"""Per-volume reconstruction (deskew -> register -> deconvolve) and its sharding over GPUs.

The unit of work is one ``(position, timepoint, channel)`` volume; units are independent (the
reference treats every (t, p) stack on its own: ``shrimpy/dynatrack/manager.py:357-384``, one job
per stack ``shrimpy/dynatrack/worker.py:223-252``; production runs them as SLURM array jobs,
``docs/data_structure.md:64``).  So the multi-GPU layout is one process per GPU, a static
round-robin of units over ranks, **no collective on the data path**: every rank reads and writes
its own positions.  ``torch.distributed`` (RCCL over xGMI when the tensors live on GPUs, gloo in the
CPU tests) carries only the timing barrier and the optional final gather of results to rank 0.
"""

from __future__ import annotations

import logging
import os
import time

from dataclasses import dataclass, field
from typing import Callable, Iterable, Sequence

import numpy as np

from .settings import DeconvolveSettings, DeskewSettings, ReconstructSettings, RegisterSettings

logger = logging.getLogger(__name__)

__all__ = [
    "Unit", "enumerate_units", "shard_units", "VolumeReconstructor", "run_sharded",
    "gather_to_rank0", "gaussian_psf_factors",
]


@dataclass(frozen=True)
class Unit:
    """One independent volume of a plate: position key (``"row/col/fov"``), timepoint, channel."""

    position: str
    t: int = 0
    c: int = 0


def enumerate_units(positions: Sequence[str], n_t: int = 1, channels: Sequence[int] = (0,)) -> list[Unit]:
    """All units of a plate in a fixed order: position-major, then time, then channel."""
    return [Unit(p, t, c) for p in positions for t in range(int(n_t)) for c in channels]


def shard_units(units: Sequence, rank: int, world_size: int) -> list:
    """Static round-robin: rank ``r`` owns units ``r, r + W, r + 2W, ...``.

    Deterministic, needs no communication, and balances to within one unit (96 positions over 8
    GPUs -> 12 each).
    """
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank/world_size {rank}/{world_size}")
    return list(units[rank::world_size])


def gaussian_psf_factors(shape_zyx=(9, 7, 7), sigma_zyx=(2.0, 1.2, 1.2)):
    """1-D factors of the separable anisotropic Gaussian PSF, each normalised to sum 1 (float32)."""
    ks = []
    for n, s in zip(shape_zyx, sigma_zyx):
        g = np.exp(-0.5 * ((np.arange(n) - n // 2) / s) ** 2)
        ks.append((g / g.sum()).astype(np.float32))
    return tuple(ks)


def _lib_error(message: str):
    from ._lib import LsrError

    return LsrError("VolumeReconstructor", -1, message)


class VolumeReconstructor:
    """(Flat-field) -> deskew -> (affine register) -> (Richardson-Lucy) for volumes of one raw shape
    on one device.

    Plans (PSF taps, border normalisation, padded working volumes) are built once and reused for
    every unit the rank owns.
    """

    def __init__(self, raw_shape_zyx, settings: ReconstructSettings, device):
        import torch

        self.device = torch.device(device)
        self.raw_shape = tuple(int(v) for v in raw_shape_zyx)
        self.settings = settings
        shape = self.raw_shape
        self._geo = None
        if settings.deskew is not None:
            from .geometry import deskew_geometry

            d: DeskewSettings = settings.deskew
            self._geo = deskew_geometry(shape, d.ls_angle_deg, d.px_to_scan_ratio, d.keep_overhang,
                                        d.average_n_slices, d.pixel_size_um)
            from .geometry import orient_shape

            shape = orient_shape(self._geo.output_shape, d.orientation)
            # un-oriented output: the deskew kernel (either border rule) can write the RL input in place
            self._canonical_deskew = d.orientation in ("identity", "")
        self._register: RegisterSettings | None = settings.registration
        self._register_matrix = None
        self.register_origin = (0, 0, 0)    # target coordinate of output index 0 (non-zero with keep_overhang)
        if self._register is not None:
            self._register_matrix, shape, self.register_origin = self._register.resolved(shape)
        self.output_shape = tuple(shape)
        self._plan = None
        self._y_pad = None
        self._pitched = None      # deskew target with zero-padded rows, when a registration reads it (see __call__)
        dec: DeconvolveSettings | None = settings.deconvolution
        self._host_rl = None      # device cpu: the arguments of host.richardson_lucy instead of a plan
        if self.device.type == "cpu":
            # no HIP device in play (the reference's cpu branch, shrimpy/preprocessing.py:78-82): every stage
            # runs its native host twin through the same public functions (shrimpy_amd/host.py)
            if dec is not None and dec.iterations > 0:
                if dec.psf_path:
                    self._host_rl = dict(psf=dec.load_psf(), separable=dec.separable,
                                         separable_rtol=dec.separable_rtol)
                else:
                    factors = gaussian_psf_factors(dec.gaussian_shape_zyx, dec.gaussian_sigma_zyx)
                    if dec.separable == "never":
                        self._host_rl = dict(psf=factors[0][:, None, None] * factors[1][None, :, None] * factors[2][None, None, :],
                                             separable="never")
                    else:
                        self._host_rl = dict(psf=None, psf_factors=factors)
        elif dec is not None and dec.iterations > 0:
            from .deconvolve import make_plan

            if dec.psf_path:
                psf = dec.load_psf()
                # (a measured PSF beyond the stencil kernels' extents runs in the Fourier domain: deconvolve_fft.py)
                self._plan = make_plan(
                    self.output_shape, psf, self.device,
                    separable={"auto": "auto", "force": "force", "never": "never"}[dec.separable],
                    separable_rtol=dec.separable_rtol, method=dec.method)
            else:
                factors = gaussian_psf_factors(dec.gaussian_shape_zyx, dec.gaussian_sigma_zyx)
                if dec.separable == "never":
                    psf = factors[0][:, None, None] * factors[1][None, :, None] * factors[2][None, None, :]
                    self._plan = make_plan(self.output_shape, psf, self.device, separable="never", method=dec.method)
                else:
                    self._plan = make_plan(self.output_shape, None, self.device, psf_factors=factors, method=dec.method)

    def __call__(self, raw, rl_events=None):
        """``raw``: (Z, Y, X) numpy array or tensor -> reconstructed float32 tensor on ``device``.
        ``rl_events``: optional ``(start, end)`` events recorded right around the RL launches."""
        import torch

        from .deskew import deskew_with_matrix
        from .register import apply_affine_transform_zyx

        # uint16 camera stacks are uploaded as they are (half the PCIe bytes; the flat-field and
        # deskew kernels convert exactly)
        keep_u16 = ((self._geo is not None or getattr(self.settings, "flatfield", False))
                    and getattr(raw, "dtype", None) in (np.uint16, np.dtype("uint16"), torch.uint16))
        if keep_u16:
            vol = torch.as_tensor(raw, device=self.device).contiguous()
        else:
            vol = torch.as_tensor(raw, device=self.device, dtype=torch.float32).contiguous()
        if tuple(vol.shape) != self.raw_shape:
            raise ValueError(f"expected raw shape {self.raw_shape}, got {tuple(vol.shape)}")
        flat = None
        if getattr(self.settings, "flatfield", False):
            from .flatfield import flat_field_pattern

            flat = flat_field_pattern(vol)
            if self._geo is None:
                vol = flat.apply(vol)
        if self._geo is not None:
            target = None
            d = self.settings.deskew
            if (self._plan is not None and self._register is None and self._plan.padded_input
                    and self._canonical_deskew):
                # deskew straight into the RL kernels' padded, line-aligned input volume
                if self._y_pad is None:
                    self._y_pad = self._plan.new_padded_input()
                target = self._y_pad
            elif (self.device.type == "cuda" and self._register is not None and self._canonical_deskew
                    and self._geo.output_shape[2] % 4 != 0 and self._geo.output_shape[2] >= 8):
                # a registration follows and the deskewed rows would not start on 16-byte boundaries:
                # deskew into zero-padded rows so that the LDS-staged affine kernels take the map
                if self._pitched is None:
                    from .register import PitchedVolume

                    self._pitched = PitchedVolume(self._geo.output_shape, self.device)
                target = self._pitched
            # (with flat-field on, its division rides along inside the deskew kernel)
            vol = deskew_with_matrix(vol, self._geo.matrix_3x4, self._geo.pre_average_shape,
                                     d.average_n_slices, out=target, flat_field=flat, border=d.border, cval=d.cval)
            if not self._canonical_deskew:
                from .deskew import orient_volume

                vol = orient_volume(vol, d.orientation)
        if self._register is not None:
            r = self._register
            target = None
            if self._plan is not None and self._plan.padded_input:
                # resample straight into the RL kernels' padded, line-aligned input volume
                if self._y_pad is None:
                    self._y_pad = self._plan.new_padded_input()
                target = self._y_pad
            vol = apply_affine_transform_zyx(vol, self._register_matrix, self.output_shape,
                                             mode=r.mode, cval=r.cval, out=target)
        if self._plan is not None:
            dec = self.settings.deconvolution
            vol = self._plan(vol, iterations=dec.iterations, eps=dec.eps, events=rl_events)
        elif self._host_rl is not None:
            from .deconvolve import richardson_lucy

            dec = self.settings.deconvolution
            vol = richardson_lucy(vol, iterations=dec.iterations, eps=dec.eps, **self._host_rl)
        return vol


@dataclass
class ShardReport:
    rank: int
    world_size: int
    units: list = field(default_factory=list)
    seconds: float = 0.0         # this rank's wall time
    max_seconds: float = 0.0     # max over ranks (what the job took)
    n_units_total: int = 0
    # seconds each stage of the staged pipeline was busy, summed over this rank's units (``_run_staged``):
    # wait_slot / load / write / collect on the two host threads, wait_load / process / wait_store on the caller
    stage_seconds: dict = field(default_factory=dict)
    # seconds between the hand-overs of consecutive units to the download / writer side (``_run_staged``): the run's pace
    # once the pipeline is full -- a short run's ``seconds / units`` also carries the first read and the last write
    unit_intervals: list = field(default_factory=list)
    # units this rank gave up on (``on_error="skip"``): (unit, stage, message), in unit order
    failures: list = field(default_factory=list)


def _dist():
    import torch.distributed as dist

    return dist if (dist.is_available() and dist.is_initialized()) else None


def run_sharded(
    units: Sequence,
    load: Callable[[object], object],
    process: Callable[[object], object],
    store: Callable[[object, object], None],
    *,
    synchronize: Callable[[], None] | None = None,
    overlap_io: bool = True,
    stager=None,
    process_takes_unit: bool = False,
    on_error: str = "raise",
) -> ShardReport:
    """Run ``store(unit, process(load(unit)))`` for this rank's share of ``units``.

    Rank and world size come from the initialised ``torch.distributed`` group (single process
    otherwise).  There is no data-path collective; a barrier brackets the timed region and the
    reported job time is the max over ranks.

    With ``overlap_io`` the host side is a three-stage pipeline: while the GPU processes unit k, one
    background thread reads unit k+1 (``load``) and another writes unit k-1 (``store``), so disk
    I/O hides behind compute (at config-2 size a volume is 8.6 GB in and 3.2 GB out against ~90 ms
    of kernels: the I/O is what needs hiding).  ``process`` always runs on the calling thread.

    With a ``stager`` (``staging.VolumeStager``) the PCIe copies leave the step as well: ``load``
    fills a pinned slot (in place when it takes ``out=``), the upload of unit k+1 and the download
    of unit k-1 run on their own HIP streams beside the kernels of unit k, ``process`` receives the
    device tensor and ``store`` the result as a (pinned) numpy array.  Events order the three
    streams; there is no device-wide synchronise inside the loop.

    ``process_takes_unit``: call ``process(data, unit)`` (the CLI picks a per-channel reconstructor).
    An exception from ``load``, ``process`` or ``store`` propagates after the worker threads have
    stopped and the stager's copy streams have drained; units finished before it stay written.

    ``on_error="skip"``: a unit whose ``load``, ``process`` or ``store`` raises an ``Exception`` is given up -- recorded in
    ``ShardReport.failures`` as (unit, stage, message) -- and the rank carries on with its next unit, the way the
    reference turns a failed stack into an ``{"type": "error"}`` record and keeps acquiring
    (``shrimpy/dynatrack/worker.py:262-271``): one damaged chunk does not end a 19 200-unit run.
    """
    if on_error not in ("raise", "skip"):
        raise ValueError("on_error must be 'raise' or 'skip'")
    skip = on_error == "skip"
    failures: list = []

    def guarded(stage, fn, unit, *args):
        """fn(*args), or -- when skipping -- (False, None) after noting the failure."""
        try:
            return True, fn(*args)
        except Exception as exc:  # noqa: BLE001 -- per-unit containment is the point
            if not skip:
                raise
            logger.error("unit %s: %s failed, skipped: %s: %s", unit, stage, type(exc).__name__, exc)
            failures.append((unit, stage, f"{type(exc).__name__}: {exc}"))
            return False, None

    import torch

    dist = _dist()
    rank = dist.get_rank() if dist else 0
    world = dist.get_world_size() if dist else 1
    mine = shard_units(list(units), rank, world)
    sync = synchronize or (lambda: None)
    if process_takes_unit:
        run = process
    else:
        def run(data, unit):
            return process(data)

    if dist:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    stage_seconds: dict = {}
    marks: list = []
    if stager is not None and mine:
        _run_staged(mine, load, run, store, stager, stage_seconds, failures if skip else None, marks)
    elif skip:
        for unit in mine:
            ok, data = guarded("load", load, unit, unit)
            if ok:
                ok, result = guarded("process", run, unit, data, unit)
            if ok:
                sync()
                guarded("store", store, unit, unit, result)
    elif overlap_io and len(mine) > 1:
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(1, "lsr-load") as loader, ThreadPoolExecutor(1, "lsr-store") as storer:
            pending_store = None
            nxt = loader.submit(load, mine[0])
            for i, unit in enumerate(mine):
                data = nxt.result()
                if i + 1 < len(mine):
                    nxt = loader.submit(load, mine[i + 1])
                result = run(data, unit)
                sync()  # the result must be complete before another thread reads it
                if pending_store is not None:
                    pending_store.result()  # surface store errors, keep at most one in flight
                pending_store = storer.submit(store, unit, result)
            if pending_store is not None:
                pending_store.result()
    else:
        for unit in mine:
            store(unit, run(load(unit), unit))
    sync()
    seconds = time.perf_counter() - t0
    max_seconds = seconds
    if dist:
        backend = dist.get_backend()
        dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        t = torch.tensor([seconds], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        max_seconds = float(t.item())
        dist.barrier()
    logger.info("rank %d/%d: %d of %d units in %.3fs (job %.3fs)", rank, world, len(mine), len(units),
                seconds, max_seconds)
    failures.sort(key=lambda f: mine.index(f[0]) if f[0] in mine else len(mine))
    return ShardReport(rank, world, mine, seconds, max_seconds, len(units), stage_seconds, failures=failures,
                       unit_intervals=[b - a for a, b in zip(marks, marks[1:])])


def _run_staged(mine, load, process, store, stager, times: dict | None = None, failures: list | None = None,
                marks: list | None = None) -> None:
    """The ``stager`` branch of ``run_sharded``: loader thread -> up stream -> kernels -> down
    stream -> writer thread, slot ``i % depth`` for the i-th unit.  ``process(data, unit)``.
    ``times`` collects how long each stage was busy (each key is touched by one thread only).
    ``failures`` (a list): skip mode -- a unit whose stage raises is appended as (unit, stage, message) and left out;
    the slots keep their order (slot ``i % depth`` belongs to unit i whether or not it got that far)."""
    import inspect

    from concurrent.futures import ThreadPoolExecutor

    try:
        takes_out = "out" in inspect.signature(load).parameters
    except (TypeError, ValueError):
        takes_out = False
    depth = stager.depth
    times = {} if times is None else times
    clock = time.perf_counter

    per_unit = [] if os.environ.get("LSR_STAGE_EVENTS") == "1" else None     # [(key, ms), ...] of the caller's clocks

    def spent(key, since):
        now = clock()
        times[key] = times.get(key, 0.0) + (now - since)
        if per_unit is not None and key in ("wait_load", "process", "wait_store", "stage_out"):
            per_unit.append((key, round((now - since) * 1e3, 1)))
        return now

    def stage(i):
        slot = i % depth
        t = clock()
        view = stager.host_in(slot)            # waits until the slot's previous upload is done
        t = spent("wait_slot", t)
        data = load(mine[i], out=view) if takes_out else load(mine[i])
        t = spent("load", t)
        slot = stager.stage_in(slot, data)
        spent("stage_in", t)
        return slot

    failed = []
    skip = failures is not None

    def note(i, stage_name, exc):
        logger.error("unit %s: %s failed, skipped: %s: %s", mine[i], stage_name, type(exc).__name__, exc)
        failures.append((mine[i], stage_name, f"{type(exc).__name__}: {exc}"))

    def collect(i):
        t = clock()
        host = stager.collect(i % depth)        # waits for the download of unit i (device-encoded results: starts it)
        spent("collect", t)
        return host

    def write(i, collected):
        if failed:                 # an earlier unit's write failed: nothing after it is written
            return
        try:
            host = collected.result()
            t = clock()
            store(mine[i], host)
            spent("write", t)
        except Exception as exc:  # noqa: BLE001
            if skip:
                # (a device decoder's verdict on the unit's input arrives with its result: that is a "load" failure)
                note(i, getattr(exc, "stage", "store"), exc)
                return
            failed.append(i)
            raise
        except BaseException:
            failed.append(i)
            raise

    # the collector runs ahead of the writer: the download of unit i (for a device-encoded result it can only be sized
    # once the frame table is on the host) proceeds while unit i - 1 is still being written
    loader, storer = ThreadPoolExecutor(1, "lsr-load"), ThreadPoolExecutor(1, "lsr-store")
    collector = ThreadPoolExecutor(1, "lsr-collect")
    stores: list = []       # one future (or None: the unit was skipped) per unit, in unit order
    # the loader works up to depth - 1 units ahead of the kernels (slot i % depth is free again once unit i - depth has
    # been uploaded and consumed: stage() waits for exactly that)
    ahead: list = []
    submitted = 0
    try:
        while submitted < min(depth - 1, len(mine)):
            ahead.append(loader.submit(stage, submitted))
            submitted += 1
        # while the first unit is being read: whatever the card does once per run (code objects of the codec kernels,
        # their LDS limits) -- on a 12-unit run that is a tenth of the wall time when the first unit has to pay it
        warm = getattr(stager, "warm_up", None)
        if warm is not None and mine:
            try:
                warm()
            except Exception as exc:  # noqa: BLE001 -- a warm-up is never worth a failed run
                logger.debug("stager warm-up skipped: %s", exc)
        for i in range(len(mine)):
            t = clock()
            slot = None
            nxt = ahead.pop(0)
            try:
                slot = nxt.result()
            except Exception as exc:  # noqa: BLE001
                if not skip:
                    raise
                note(i, "load", exc)
            t = spent("wait_load", t)
            if submitted < len(mine):
                ahead.append(loader.submit(stage, submitted))
                submitted += 1
            result = None
            if slot is not None:
                acquired = False
                try:
                    data = stager.acquire(slot)
                    acquired = True
                    result = process(data, mine[i])
                except Exception as exc:  # noqa: BLE001
                    if not skip:
                        raise
                    note(i, "process" if acquired else "load", exc)   # (a chunk the device decoder refuses shows up at acquire)
                    result = None
                finally:
                    stager.release(slot)
            t = spent("process", t)
            # the result slot of unit i was last used by unit i - depth: its write must be over before
            # the download of unit i lands there.  The write of unit i - 1 may still be running -- it
            # overlaps the kernels and the download of unit i (waiting for it here instead cost a
            # third of the streamed rate: 0.11 s per config-4 unit against 0.05 s of the slowest stage)
            if i - depth >= 0 and stores[i - depth] is not None:
                stores[i - depth].result()
            t = spent("wait_store", t)
            if result is None:
                stores.append(None)
                continue
            stager.stage_out(i % depth, result)
            spent("stage_out", t)
            if marks is not None:
                marks.append(clock())
            stores.append(storer.submit(write, i, collector.submit(collect, i)))
        t = clock()
        for fut in stores:
            if fut is not None:
                fut.result()        # the last writes; raises the first writer error, if any
        spent("wait_store", t)
        stores = []
    finally:
        # whatever happened: no thread is left filling a slot, no copy is left in flight on the
        # up / down streams, before the caller sees the exception (or the result)
        # (a queued load is dropped; a queued write belongs to a finished unit and is completed)
        for fut in ahead:
            if not fut.cancel():
                try:
                    fut.result()
                except Exception:  # noqa: BLE001 -- the first failure is the one that propagates
                    pass
        for fut in stores:
            try:
                if fut is not None:
                    fut.result()
            except Exception:  # noqa: BLE001
                pass
        loader.shutdown(wait=True)
        collector.shutdown(wait=True)
        storer.shutdown(wait=True)
        stager.drain()
        if per_unit:
            logger.warning("caller's clocks in order, ms: %s", per_unit)


def gather_to_rank0(local: Iterable, n_total: int, keep_on_host: bool = False):
    """Optional final gather (the north-star's "stitched-volume gather"): rank 0 receives every
    rank's result tensors in unit order; other ranks return ``None``.

    ``local`` = this rank's results in the order of ``shard_units`` (units ``rank, rank + W, ...``); ranks may own
    different numbers of units and units may differ in shape -- every rank first publishes (shape, dtype) of what it
    holds.  The transfer goes round by round: in round k rank 0 posts the receives of EVERY peer's k-th unit at once
    (``batch_isend_irecv``: one grouped launch under RCCL), so the W - 1 incoming transfers of a round run side by side,
    each over its own xGMI link -- a point-to-point gather, not a ring all-gather, which one link would bound for this
    many-to-one pattern.  At most W - 1 received volumes are in flight on rank 0's device at a time.

    ``keep_on_host``: rank 0 returns CPU tensors (each round's volumes are copied to host memory before the next round is
    received): 96 results of 1.75 GB (config 4) do not have to fit rank 0's HBM next to its own working set.
    """
    import torch

    dist = _dist()
    local = list(local)
    if dist is None:
        return [t.cpu() if keep_on_host and hasattr(t, "cpu") else t for t in local]
    rank, world = dist.get_rank(), dist.get_world_size()
    metas: list = [None] * world
    dist.all_gather_object(metas, [(tuple(t.shape), str(t.dtype).replace("torch.", "")) for t in local])
    counts = [len(m) for m in metas]
    expect = [len(range(r, n_total, world)) for r in range(world)]
    if counts != expect:
        raise RuntimeError(f"ranks hold {counts} results, a round-robin of {n_total} units over {world} ranks is {expect}")
    # gloo moves host memory only: device tensors are staged through the host (ranks sharing one card in a
    # rehearsal, or a CPU-only box); RCCL sends them GPU to GPU
    via_host = dist.get_backend() == "gloo" and any(t.is_cuda for t in local)
    rounds = max(counts) if counts else 0
    if rank != 0:
        for k in range(rounds):
            if k < len(local):
                t = local[k].contiguous()
                for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, t.cpu() if via_host else t, 0)]):
                    req.wait()
        return None
    device = local[0].device if local else torch.device("cpu")
    recv_device = torch.device("cpu") if (via_host or dist.get_backend() == "gloo") else device
    out: list = [None] * n_total
    for i, t in enumerate(local):
        out[i * world] = t.cpu() if keep_on_host else t
    for k in range(rounds):
        ops, slots = [], []
        for src in range(1, world):
            if k < counts[src]:
                shape, dtype = metas[src][k]
                buf = torch.empty(shape, dtype=getattr(torch, dtype), device=recv_device)
                ops.append(dist.P2POp(dist.irecv, buf, src))
                slots.append((src + k * world, buf))
        if not ops:
            continue
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        for idx, buf in slots:
            out[idx] = buf.cpu() if keep_on_host else (buf.to(device) if buf.device != device else buf)
    return out
