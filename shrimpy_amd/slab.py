"""One volume over several GPUs: row slabs with a halo exchange per RL iteration (SURVEY 8 e / f-4).

The per-unit sharding of ``pipeline.py`` scales a plate; this module scales ONE volume when it has
to go faster than one GPU runs it.  The split is along Y' (the deskewed row axis):

* the deskew is halo-free along it -- Y' is raw X reversed, one to one (the reference chunks the
  same way, ``scripts/measure_psf.py:221-249``) -- so a rank deskews its own raw-X slab straight
  into its slab of the deskewed volume (:func:`deskew_slab`);
* a fused RL iteration (``x_new = x * H^T(y / (H x)) / H^T 1``) reaches two in-plane PSF radii, so
  a rank keeps ``halo = 2 * (py // 2)`` extra rows either side of the rows it owns, runs the
  ordinary single-GPU kernel on that extended slab (with the TALL volume's border normalisation,
  ``RichardsonLucyPlan(y_window=...)``), and after every iteration replaces its halo rows by the
  neighbours' freshly computed ones: ``halo * Z * X * 4`` bytes per neighbour and iteration (9.3 MB
  at config 2), point to point over xGMI -- no collective.  Owned rows then equal the
  single-GPU result bit for bit: each voxel sees the same inputs and the same arithmetic.

``exchange`` is pluggable: :func:`exchange_halos` uses ``torch.distributed`` send / recv (RCCL on a
GPU node; staged through the host for the ``gloo`` backend of the CPU tests); the single-process
emulation in the parity tests copies between the slabs directly.
"""

from __future__ import annotations

from dataclasses import dataclass

__all__ = ["Slab", "slab_ranges", "SlabRichardsonLucy", "exchange_halos", "deskew_slab", "run_slabs_in_process"]


@dataclass(frozen=True)
class Slab:
    """Rows ``[own0, own1)`` are this rank's; it holds ``[ext0, ext1)`` (own rows + halos)."""

    rank: int
    world: int
    own0: int
    own1: int
    ext0: int
    ext1: int

    @property
    def lo(self) -> int:   # halo rows below the owned ones
        return self.own0 - self.ext0

    @property
    def hi(self) -> int:
        return self.ext1 - self.own1

    @property
    def rows(self) -> int:
        return self.ext1 - self.ext0


def slab_ranges(n_rows: int, world: int, halo: int) -> list[Slab]:
    """Balanced contiguous row ranges with ``halo`` extra rows either side (clipped at the volume)."""
    n_rows, world, halo = int(n_rows), int(world), int(halo)
    if world < 1 or n_rows < world:
        raise ValueError(f"cannot split {n_rows} rows over {world} ranks")
    base, extra = divmod(n_rows, world)
    out, start = [], 0
    for r in range(world):
        stop = start + base + (1 if r < extra else 0)
        if world > 1 and stop - start < halo:
            raise ValueError(f"slabs of {stop - start} rows are thinner than the halo ({halo}): use fewer ranks")
        out.append(Slab(r, world, start, stop, max(0, start - halo), min(n_rows, stop + halo)))
        start = stop
    return out


def exchange_halos(view, slab: Slab, group=None) -> None:
    """Refresh the halo rows of ``view`` (``(Z, slab.rows, X)``, the logical window of a padded
    volume) from the neighbouring ranks' owned rows.  Point-to-point, both directions at once."""
    import torch
    import torch.distributed as dist

    if slab.world == 1:
        return
    via_host = dist.get_backend(group) == "gloo" and view.is_cuda
    ops, recvs = [], []

    def staged(t):
        return t.contiguous().cpu() if via_host else t.contiguous()

    lo, hi, n = slab.lo, slab.hi, slab.rows
    if slab.rank > 0:  # lower neighbour: it needs my first `its hi` owned rows; I need its last `lo`
        send = staged(view[:, lo:lo + _neighbour_halo(slab, -1), :])
        recv = torch.empty((view.shape[0], lo, view.shape[2]), dtype=view.dtype, device=send.device)
        ops += [dist.P2POp(dist.isend, send, slab.rank - 1, group), dist.P2POp(dist.irecv, recv, slab.rank - 1, group)]
        recvs.append((slice(0, lo), recv))
    if slab.rank < slab.world - 1:
        send = staged(view[:, n - hi - _neighbour_halo(slab, +1):n - hi, :])
        recv = torch.empty((view.shape[0], hi, view.shape[2]), dtype=view.dtype, device=send.device)
        ops += [dist.P2POp(dist.isend, send, slab.rank + 1, group), dist.P2POp(dist.irecv, recv, slab.rank + 1, group)]
        recvs.append((slice(n - hi, n), recv))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    for rows, buf in recvs:
        view[:, rows, :].copy_(buf.to(view.device))


def _neighbour_halo(slab: Slab, direction: int) -> int:
    """Rows the neighbour in ``direction`` keeps as halo on the side facing this rank: the full
    halo, which is also this rank's own halo on that side (only the volume's borders clip one)."""
    return slab.own0 - slab.ext0 if direction < 0 else slab.ext1 - slab.own1


def run_slabs_in_process(slab_rls: "list[SlabRichardsonLucy]", iterations: int = 20, eps: float = 1e-6):
    """All ranks of a split in ONE process, in lockstep, halos copied directly between the slabs --
    the reference behaviour of the distributed run (used by the parity tests, which have one GPU).
    Returns the owned rows of every slab, in rank order."""
    src = [s._x[0] for s in slab_rls]
    dst = [s._x[1] for s in slab_rls]
    for s, v in zip(slab_rls, src):
        v.view.copy_(s.y_pad.view)
    for it in range(int(iterations)):
        for s, a, b in zip(slab_rls, src, dst):
            s.plan.iterate_padded(s.y_pad, a, b, eps=eps)
        if it + 1 < iterations:
            for r, s in enumerate(slab_rls):
                sl, n = s.slab, s.slab.rows
                if r > 0:      # my lower halo <- the last rows the lower neighbour owns
                    nb = slab_rls[r - 1].slab
                    dst[r].view[:, 0:sl.lo, :].copy_(dst[r - 1].view[:, nb.rows - nb.hi - sl.lo:nb.rows - nb.hi, :])
                if r + 1 < len(slab_rls):
                    nb = slab_rls[r + 1].slab
                    dst[r].view[:, n - sl.hi:n, :].copy_(dst[r + 1].view[:, nb.lo:nb.lo + sl.hi, :])
        src, dst = dst, src
    return [v.view[:, s.slab.lo:s.slab.lo + (s.slab.own1 - s.slab.own0), :] for s, v in zip(slab_rls, src)]


class SlabRichardsonLucy:
    """Richardson-Lucy on this rank's row slab of a ``shape_zyx`` volume (separable PSF, fused path)."""

    def __init__(self, shape_zyx, psf_factors, device, rank: int, world: int):
        from .deconvolve import PaddedVolume, RichardsonLucyPlan

        z, y, x = (int(v) for v in shape_zyx)
        py = len(psf_factors[1])
        self.halo = 2 * (py // 2)
        self.slab = slab_ranges(y, world, self.halo)[rank]
        self.full_shape = (z, y, x)
        self.plan = RichardsonLucyPlan((z, self.slab.rows, x), None, device, psf_factors=psf_factors,
                                       y_window=(self.slab.ext0, y))
        if not self.plan.fused:
            raise ValueError("the slab split runs the fused separable kernel; this PSF has no specialisation")
        self.y_pad = self.plan.new_padded_input()      # the caller (or deskew_slab) fills .view
        self._x = [PaddedVolume(self.plan.shape, self.plan._psf.shape, device) for _ in range(2)]

    def run(self, iterations: int = 20, eps: float = 1e-6, exchange=None):
        """RL from ``x0 = y`` on the extended slab; returns the OWNED rows ``(Z, own rows, X)`` (a view).

        ``exchange(view, slab)`` refreshes halo rows after each iteration (default:
        :func:`exchange_halos` over the default process group)."""
        if exchange is None:
            exchange = exchange_halos
        src, dst = self._x
        src.view.copy_(self.y_pad.view)
        for it in range(int(iterations)):
            self.plan.iterate_padded(self.y_pad, src, dst, eps=eps)
            if it + 1 < iterations:
                exchange(dst.view, self.slab)
            src, dst = dst, src
        s = self.slab
        return src.view[:, s.lo:s.lo + (s.own1 - s.own0), :]


def deskew_slab(raw_x_slab, slab_rl: SlabRichardsonLucy, ls_angle_deg: float, px_to_scan_ratio: float,
                keep_overhang: bool, average_n_slices: int = 1, flat_field=None):
    """Deskew the raw-X slab that maps onto this rank's extended rows, straight into the slab's
    padded RL input.  ``raw_x_slab`` = ``raw[:, :, X - ext1 : X - ext0]`` (Y' is raw X reversed)."""
    from .deskew import deskew_with_matrix
    from .geometry import deskew_geometry

    geo = deskew_geometry(tuple(raw_x_slab.shape), ls_angle_deg, px_to_scan_ratio, keep_overhang, average_n_slices)
    if tuple(geo.output_shape) != slab_rl.plan.shape:
        raise ValueError(f"raw slab deskews to {tuple(geo.output_shape)}, the slab is {slab_rl.plan.shape}")
    return deskew_with_matrix(raw_x_slab, geo.matrix_3x4, geo.pre_average_shape, average_n_slices,
                              out=slab_rl.y_pad, flat_field=flat_field)
