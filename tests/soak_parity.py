"""Randomised parity soak: many seeded random cases per kernel against the oracle (or against the
other device path where the bar is bit-equality), for a bounded time.  Not part of the test suite --
a one-off sweep for shapes and parameters nobody wrote a case for (it lives under tests/ because it
uses the oracle as its checker; pytest does not collect it).  Prints one JSON line per family
(cases run, failures with their seeds) and exits non-zero if anything failed.

    python tests/soak_parity.py --seconds 240 --seed 1
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import cpu_ref as o  # noqa: E402  (a checker, like the tests)


def main():
    import torch
    from scipy import ndimage

    from shrimpy_amd import _lib
    from shrimpy_amd import dynatrack as d
    from shrimpy_amd.deconvolve import RichardsonLucyPlan
    from shrimpy_amd.deskew import fast_deskew_zyx
    from shrimpy_amd.flatfield import flat_field_pattern
    from shrimpy_amd.deconvolve import PaddedVolume
    from shrimpy_amd.register import PitchedVolume, apply_affine_transform_zyx

    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", default="", help="comma-separated family names")
    ap.add_argument("--large", action="store_true", help="large-shape families only (work split, big grids)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(args.seed)

    def t(a):
        return torch.as_tensor(np.ascontiguousarray(a), device=dev)

    def deskew_case(r):
        shape = (int(r.integers(2, 160)), int(r.integers(1, 48)), int(r.integers(1, 90)))
        angle = float(r.uniform(8.0, 45.0))
        ratio = float(np.round(r.uniform(0.25, 2.0), 3))
        keep, avg = bool(r.integers(0, 2)), int(r.integers(1, 6))
        u16 = bool(r.integers(0, 2))
        border = "grid-constant" if r.integers(0, 2) else "constant"     # both rules run the fused kernel
        try:
            want_shape = o.deskewed_shape(shape, angle, ratio, keep, avg)[0]
        except Exception:
            return None
        if min(want_shape) <= 0:
            return None
        raw = r.integers(90, 60000, shape).astype(np.uint16) if u16 else (r.random(shape) * 4000 - 500).astype(np.float32)
        cval = [0.0, "min", float(np.round(r.uniform(-50, 200), 1))][int(r.integers(0, 3))]     # (round 5: the fill value)
        want = o.deskew(raw.astype(np.float32), angle, ratio, keep, avg, border=border, cval=cval)
        got = fast_deskew_zyx(raw_data=t(raw), ls_angle_deg=angle, px_to_scan_ratio=ratio, keep_overhang=keep,
                              average_n_slices=avg, border=border, cval=cval).cpu().numpy()
        return got.shape == want.shape and np.array_equal(got, want), (shape, angle, ratio, keep, avg, u16, border, cval)

    def affine_case(r):
        shape = (int(r.integers(1, 40)), int(r.integers(2, 90)), int(r.integers(2, 140)))
        if r.integers(0, 3):   # the LDS-staged kernels (planar, box) need a row length that is a multiple of 4
            shape = shape[:2] + (4 * int(r.integers(2, 36)),)
        oshape = shape if r.integers(0, 2) else (int(r.integers(1, 24)), int(r.integers(2, 90)), int(r.integers(2, 140)))
        th = np.deg2rad(r.uniform(-25, 25))
        m = np.eye(4)
        planar = bool(r.integers(0, 2))
        rot = np.array([[1, 0, 0], [0, np.cos(th), -np.sin(th)], [0, np.sin(th), np.cos(th)]])
        m[:3, :3] = rot @ np.diag(r.uniform(0.6, 1.5, 3) * r.choice([1.0, 1.0, -1.0], 3))
        if not planar:
            m[0, 1:3] = r.uniform(-0.2, 0.2, 2)
            m[1:3, 0] = r.uniform(-0.2, 0.2, 2)
        m[:3, 3] = r.uniform(-10, 10, 3) + np.where(m.diagonal()[:3] < 0, np.array(shape) - 1.0, 0.0)
        mode = "constant" if r.integers(0, 3) else "grid-constant"
        vol = (r.random(shape) * 1000 - 100).astype(np.float32)
        want = o.affine_apply_4x4(vol, m, oshape, cval=-3.0, mode=mode)
        # any width through zero-padded rows (the LDS kernels then take it), any destination strides
        src = PitchedVolume.copy_of(t(vol)) if r.integers(0, 2) else t(vol)
        dst = None
        if r.integers(0, 2):
            dst = PaddedVolume(oshape, (int(r.choice([3, 5, 9])), int(r.choice([3, 7])), int(r.choice([3, 7]))), dev)
        got = apply_affine_transform_zyx(src, m, oshape, mode=mode, cval=-3.0, out=dst)
        got = (got.view if dst is not None else got).cpu().numpy()
        ok = np.array_equal(got, want)
        if ok and dst is not None:     # nothing outside the window was touched
            dst.view.zero_()
            ok = not bool(dst.full.any())
        if ok and mode == "constant":   # f32 interpolation: same border decisions, close values
            g32 = apply_affine_transform_zyx(t(vol), m, oshape, cval=-3.0, exact=False).cpu().numpy()
            ok = np.array_equal(g32 == -3.0, want == -3.0) and float(np.abs(g32 - want).max()) <= 2e-5 * 1100
        return ok, (shape, oshape, mode, planar, type(src).__name__, dst is not None, m[:3].round(4).tolist())

    odd = np.array([1, 3, 5, 7, 9, 11, 13, 15])

    def rl_case(r):
        pshape = tuple(int(v) for v in r.choice(odd, 3))
        if pshape[0] >= 15 and max(pshape[1:]) >= 11:
            pshape = (13,) + pshape[1:]
        vshape = (int(r.integers(1, 48)), int(r.integers(1, 100)), int(r.integers(1, 300)))
        factors = [np.abs(r.normal(1.0, 0.4, n)).astype(np.float32) + 0.05 for n in pshape]
        factors = [f / f.sum() for f in factors]
        y = t((r.random(vshape) * 80 + 1).astype(np.float32))
        iters = int(r.integers(1, 4))
        a = RichardsonLucyPlan(vshape, None, dev, psf_factors=factors, fused="always")(y, iterations=iters)
        b = RichardsonLucyPlan(vshape, None, dev, psf_factors=factors, fused="never")(y, iterations=iters)
        ok = bool(torch.equal(a, b))
        if ok and np.prod(vshape) < 200000 and max(pshape) <= 9:   # and against the oracle, where it is quick
            want = o.richardson_lucy_separable(y.cpu().numpy(), factors, iterations=iters).astype(np.float64)
            got = a.cpu().numpy().astype(np.float64)
            ok = bool(np.all(np.abs(got - want) <= 2e-4 * np.abs(want) + 1e-4 * np.abs(want).max()))
        return ok, (pshape, vshape, iters)

    def flat_case(r):
        shape = (int(r.integers(1, 300)), int(r.integers(1, 20)), int(r.integers(1, 200)))
        kind = int(r.integers(0, 3))
        if kind == 0:
            vol = r.integers(0, 40, shape).astype(np.float32)          # counts with many ties
        elif kind == 1:
            vol = r.integers(0, 65535, shape).astype(np.uint16)
        else:
            vol = (r.normal(0, 50, shape)).astype(np.float32)
        srt = np.sort(vol.astype(np.float32), axis=0)
        n = shape[0]
        a, b = srt[(n - 1) // 2], srt[n // 2]
        want = b - (b - a) * np.float32(0.5)
        got = flat_field_pattern(t(vol)).pattern.cpu().numpy()
        return np.array_equal(got, want), (shape, kind)

    def blur_case(r):
        shape = (int(r.integers(1, 120)), int(r.integers(1, 70)), int(r.integers(1, 300)))
        axis = int(r.integers(0, 3))
        rad = int(r.integers(0, min(shape[axis], 40)))
        vol = (r.random(shape) * 900 + 100).astype(np.float32)
        taps = r.random(2 * rad + 1).astype(np.float32)
        taps /= taps.sum()
        want = ndimage.correlate1d(vol.astype(np.float64), taps.astype(np.float64), axis=axis, mode="mirror")
        src, out, dt = t(vol), torch.empty(shape, dtype=torch.float32, device=dev), t(taps)
        _lib.call("lsr_blur_reflect_f32", src.data_ptr(), out.data_ptr(), *shape, axis, dt.data_ptr(), rad,
                  ctypes.c_float(0.0), ctypes.c_float(0.0), _lib.stream_ptr(dev))
        return bool(np.allclose(out.cpu().numpy(), want, rtol=2e-6, atol=2e-4)), (shape, axis, rad)

    def estimator_case(r):
        shape = (int(r.integers(1, 40)), int(r.integers(1, 60)), int(r.integers(1, 120)))
        vol = (r.gamma(2.0, 100.0, shape)).astype(np.float32)
        p = float(r.uniform(1, 99.9))
        ok = abs(d._percentile(t(vol), p) - o.dt_percentile(vol, p)) <= 1e-5 * abs(o.dt_percentile(vol, p)) + 1e-6
        bg = float(r.uniform(0, 300))
        ok = ok and np.allclose(d._intensity_center_of_mass(t(vol), bg).cpu().numpy(),
                                o.dt_intensity_center_of_mass(vol, bg), atol=2e-3)
        shift = tuple(int(v) for v in (r.integers(-2, 3), r.integers(-5, 6), r.integers(-5, 6)))
        if min(shape) >= 8:
            mov = np.roll(vol, shift, axis=(0, 1, 2))
            ok = ok and d._phase_cross_corr(t(vol), t(mov)) == o.dt_phase_cross_corr(vol, mov)
        return bool(ok), (shape, p, bg, shift)

    def rl_ysep_case(r):
        """PSFs that separate along y only: one-launch and four-launch forms against the oracle."""
        pz, py, px = int(r.choice([3, 5, 7, 9, 11])), int(r.choice([3, 5, 7, 9, 11, 15])), int(r.choice([3, 5, 7, 9]))
        kzx = np.abs(r.normal(1.0, 0.5, (pz, px))) + 0.05
        ky = np.abs(r.normal(1.0, 0.4, py)) + 0.05
        psf = (ky[None, :, None] * kzx[:, None, :]).astype(np.float32)
        psf /= psf.sum()
        shape = (int(r.integers(1, 24)), int(r.integers(1, 90)), int(r.integers(1, 160)))
        y = (r.random(shape) * 80 + 1).astype(np.float32)
        plan = RichardsonLucyPlan(shape, psf, dev)      # the default: one launch per iteration where it is compiled
        if not plan.path.startswith("y-separable"):
            return False, (psf.shape, shape, plan.path)
        iters = int(r.integers(1, 4))
        dev_out = plan(t(y), iterations=iters)
        got = dev_out.cpu().numpy().astype(np.float64)
        want = o.richardson_lucy(y, psf, iters).astype(np.float64)
        ok = bool(np.all(np.abs(got - want) <= 2e-4 * np.abs(want) + 1e-4 * np.abs(want).max()))
        if ok and py <= 9:     # ... which must carry the two-launch form's bits
            two = RichardsonLucyPlan(shape, psf, dev, fused="never")
            ok = plan.path == "y-separable (fused)" and two.path == "y-separable" and bool(torch.equal(two(t(y), iterations=iters), dev_out))
        return ok, (psf.shape, shape, iters, plan.path)

    def rl_stats_case(r):
        """The RL scalars of the update epilogues (flux, change, total) on a random path against torch's fp64 reductions
        of consecutive estimates; asking for them must not change the estimate."""
        kind = int(r.integers(0, 4))
        vshape = (int(r.integers(1, 40)), int(r.integers(1, 90)), int(r.integers(1, 260)))
        y = t((r.random(vshape) * 80 + 1).astype(np.float32))
        if kind <= 1:       # separable: one launch / pair
            pshape = tuple(int(v) for v in r.choice(odd, 3))
            if pshape[0] >= 15 and max(pshape[1:]) >= 11:
                pshape = (13,) + pshape[1:]
            factors = [np.abs(r.normal(1.0, 0.4, n)).astype(np.float32) + 0.05 for n in pshape]
            plan = RichardsonLucyPlan(vshape, None, dev, psf_factors=[f / f.sum() for f in factors],
                                      fused="always" if kind == 0 else "never")
        else:               # ky (x) kzx: one launch / pair
            pz, py, px = int(r.choice([3, 5, 7, 9, 11])), int(r.choice([3, 5, 7, 9])), int(r.choice([3, 5, 7, 9]))
            psf = ((np.abs(r.normal(1.0, 0.4, py)) + 0.05)[None, :, None] * (np.abs(r.normal(1.0, 0.5, (pz, px))) + 0.05)[:, None, :])
            psf = (psf / psf.sum()).astype(np.float32)
            plan = RichardsonLucyPlan(vshape, psf, dev, fused="auto" if kind == 2 else "never")
        xs = [y] + [plan(y, iterations=n) for n in (1, 2, 3)]
        x3 = plan(y, iterations=3, stats=True)
        s_ = plan.last_stats
        ok = bool(torch.equal(x3, xs[3]))
        for n in range(3):
            tot = float(xs[n + 1].double().sum())
            chg = float((xs[n + 1].double() - xs[n].double()).abs().sum())
            ok = ok and abs(s_.total[n] / tot - 1) < 1e-5 and (chg == 0 or abs(s_.change[n] / chg - 1) < 1e-5)
        ok = ok and bool(np.all(np.abs(s_.flux / float(y.double().sum()) - 1) < 1e-5))
        return ok, (kind, plan.path, vshape)

    def rl_long_z_case(r):
        """Separable PSFs with 17 .. 31 z taps (in-plane launch + z march) against the oracle."""
        pshape = (int(r.choice([17, 19, 21, 23, 25, 27, 29, 31])), int(r.choice([1, 3, 5, 9, 15])), int(r.choice([1, 3, 7, 11])))
        vshape = (int(r.integers(1, 70)), int(r.integers(1, 50)), int(r.integers(1, 140)))
        factors = [np.abs(r.normal(1.0, 0.4, n)).astype(np.float32) + 0.05 for n in pshape]
        factors = [f / f.sum() for f in factors]
        y = (r.random(vshape) * 80 + 1).astype(np.float32)
        plan = RichardsonLucyPlan(vshape, None, dev, psf_factors=factors)
        iters = int(r.integers(1, 4))
        got = plan(t(y), iterations=iters).cpu().numpy().astype(np.float64)
        want = o.richardson_lucy_separable(y, factors, iterations=iters).astype(np.float64)
        ok = plan.path.startswith("separable (long z") and bool(np.all(np.abs(got - want) <= 2e-4 * np.abs(want) + 1e-4 * np.abs(want).max()))
        return ok, (pshape, vshape, iters)

    def rl_fft_case(r):
        """Dense random PSFs of any odd extents through the Fourier-domain iteration against the oracle's direct stencil,
        reduction scalars included; volumes thinner than the PSF, odd widths."""
        from shrimpy_amd.deconvolve import make_plan

        pshape = tuple(int(v) for v in r.choice([1, 3, 5, 7, 9, 11, 13, 17, 21, 25], 3))
        vshape = (int(r.integers(1, 40)), int(r.integers(1, 50)), int(r.integers(1, 120)))
        w = np.abs(r.normal(1.0, 0.5, pshape)).astype(np.float32) + 0.02
        w /= w.sum()
        y = (r.random(vshape) * 80 + 1).astype(np.float32)
        plan = make_plan(vshape, w, dev, method="fft")
        iters = int(r.integers(1, 4))
        got = plan(t(y), iterations=iters, stats=True).cpu().numpy().astype(np.float64)
        want = o.richardson_lucy(y, w, iterations=iters).astype(np.float64)
        ok = plan.path == "fft" and bool(np.all(np.abs(got - want) <= 2e-4 * np.abs(want) + 1e-4 * np.abs(want).max()))
        ref = o.rl_iteration_scalars(y, w, iters)
        # (at a fixed point -- a one-tap PSF, a volume one voxel thick along the PSF's only long axis -- the true change is
        # 0 and the transforms' rounding, 1e-7 of every voxel, is all there is to sum: an absolute floor of 2e-6 of the total)
        floor = 2e-6 * float(np.max(ref["total"]))
        ok = ok and all(np.allclose(getattr(plan.last_stats, k), ref[k], rtol=2e-5, atol=floor) for k in ("flux", "change", "total"))
        return ok, (pshape, vshape, iters)

    def host_twin_case(r):
        """CPU tensors through the public functions (csrc/host_twins.hip) against the device kernels: same bits."""
        from shrimpy_amd.flatfield import flat_field_pattern as ffp

        shape = (int(r.integers(2, 120)), int(r.integers(1, 24)), int(r.integers(1, 60)))
        angle, ratio = float(r.uniform(10.0, 45.0)), float(np.round(r.uniform(0.3, 1.8), 3))
        keep, avg = bool(r.integers(0, 2)), int(r.integers(1, 5))
        border = "grid-constant" if r.integers(0, 2) else "constant"
        if min(o.deskewed_shape(shape, angle, ratio, keep, avg)[0]) <= 0:
            return None
        raw = r.integers(0, 60000, shape).astype(np.uint16) if r.integers(0, 2) else (r.random(shape) * 4000 - 500).astype(np.float32)
        kw = dict(ls_angle_deg=angle, px_to_scan_ratio=ratio, keep_overhang=keep, average_n_slices=avg, border=border)
        host_raw = torch.as_tensor(raw)
        a, b = fast_deskew_zyx(raw_data=host_raw, **kw), fast_deskew_zyx(raw_data=t(raw), **kw)
        ok = a.device.type == "cpu" and bool(torch.equal(a, b.cpu()))
        m = np.eye(4)
        m[:3, :3] += r.normal(0, 0.05, (3, 3))
        m[:3, 3] = r.uniform(-4, 4, 3)
        mode = "grid-constant" if r.integers(0, 2) else "constant"
        ok = ok and bool(torch.equal(apply_affine_transform_zyx(a, m, mode=mode, cval=1.5),
                                     apply_affine_transform_zyx(b, m, mode=mode, cval=1.5).cpu()))
        ok = ok and bool(torch.equal(ffp(host_raw).pattern, ffp(t(raw)).pattern.cpu()))
        return ok, (shape, angle, ratio, keep, avg, border, mode)

    def rl_large_case(r):
        """More tiles than CUs: the whole-column / z-piece work split of the fused kernel."""
        pshape = tuple(int(v) for v in r.choice(odd[:6], 3))
        vshape = (int(r.integers(1, 70)), int(r.integers(150, 1400)), int(r.integers(400, 2600)))
        factors = [np.abs(r.normal(1.0, 0.4, n)).astype(np.float32) + 0.05 for n in pshape]
        factors = [f / f.sum() for f in factors]
        g = torch.Generator(device=dev).manual_seed(int(r.integers(0, 2**31)))
        y = torch.rand(vshape, device=dev, generator=g) * 80 + 1
        iters = int(r.integers(1, 3))
        pa = RichardsonLucyPlan(vshape, None, dev, psf_factors=factors, fused="always")
        pb = RichardsonLucyPlan(vshape, None, dev, psf_factors=factors, fused="never")
        ok = bool(torch.equal(pa(y, iterations=iters), pb(y, iterations=iters)))
        pa.release(), pb.release()
        return ok, (pshape, vshape, iters)

    def deskew_large_case(r):
        """Bigger stacks: uint16 path against float path, and the oracle on a raw-X slab."""
        shape = (int(r.integers(100, 900)), int(r.integers(20, 300)), int(r.integers(200, 1500)))
        angle, ratio = float(r.uniform(15.0, 45.0)), float(np.round(r.uniform(0.4, 1.5), 3))
        keep, avg = bool(r.integers(0, 2)), int(r.integers(1, 5))
        if min(o.deskewed_shape(shape, angle, ratio, keep, avg)[0]) <= 0:
            return None
        raw = torch.randint(0, 60000, shape, device=dev, generator=torch.Generator(device=dev).manual_seed(
            int(r.integers(0, 2**31)))).to(torch.uint16)
        kw = dict(ls_angle_deg=angle, px_to_scan_ratio=ratio, keep_overhang=keep, average_n_slices=avg)
        a = fast_deskew_zyx(raw_data=raw, **kw)
        b = fast_deskew_zyx(raw_data=raw.to(torch.float32), **kw)
        ok = bool(torch.equal(a, b))
        x0 = int(r.integers(0, shape[2] - 2))
        want = o.deskew(raw[:, :, x0:x0 + 2].to(torch.float32).cpu().numpy(), angle, ratio, keep, avg)
        ok = ok and np.array_equal(a[:, shape[2] - x0 - 2:shape[2] - x0, :].cpu().numpy(), want)
        return ok, (shape, angle, ratio, keep, avg, x0)

    def codec_case(r):
        """Device encoder == its host twin byte for byte and readable by libzstd; device decoder on those frames and on
        libzstd's own (round 5, io/device_codec.py)."""
        from shrimpy_amd.io import codecs
        from shrimpy_amd.io.device_codec import DeviceBloscDecoder, DeviceBloscEncoder, encode_frames_host, frame_layout

        dtype = [np.uint8, np.uint16, np.float32][int(r.integers(0, 3))]
        item = np.dtype(dtype).itemsize
        n = int(r.choice([1, 300, 70000, 400000, int(r.integers(1, 900000))]))
        kind = int(r.integers(0, 4))
        v = (100 + r.poisson(r.choice([2.0, 40.0, 900.0]), n) if kind == 0 else r.integers(0, 256, n) if kind == 1
             else np.repeat(r.integers(0, 3, n // 50 + 1) * 700, 50)[:n] if kind == 2 else (1 + np.sin(np.linspace(0, 40, n))) * 3000)
        vol = v.astype(dtype) if np.dtype(dtype).kind == "f" else np.clip(v, 0, np.iinfo(dtype).max).astype(dtype)
        raw = vol.view(np.uint8).reshape(-1)
        fb = int(r.choice([4096, 65536, 1 << 20, vol.nbytes])) // item * item or item
        bs = int(r.choice([0, 4096, 32768, 65536 * item]))
        frames = DeviceBloscEncoder(vol.nbytes, item, fb, dev, bs).encode_to_host(t(vol))
        ok = frames == encode_frames_host(vol, fb, bs)
        back = np.concatenate([codecs.blosc_decode(f, backend="python") for f in frames[:3]])
        m = min(back.size, raw.size)
        ok = ok and np.array_equal(back[:m], raw[:m])
        lay = frame_layout(frames[0])
        out = torch.empty(len(frames) * fb, dtype=torch.uint8, device=dev)
        DeviceBloscDecoder(len(frames) * fb, fb, lay["blocksize"], item, dev).decode_from_host(frames, out)
        ok = ok and np.array_equal(out.cpu().numpy()[:raw.size], raw)
        own = codecs.blosc_encode(raw[:min(raw.size, 300000 * item)], item, cname="zstd", clevel=int(r.integers(1, 10)),
                                  shuffle=1 if item > 1 else 0, blocksize=int(r.choice([0, 4096, 32768])))
        lay = frame_layout(own)
        if lay is not None:
            out = torch.empty(lay["nbytes"], dtype=torch.uint8, device=dev)
            DeviceBloscDecoder(lay["nbytes"], lay["nbytes"], lay["blocksize"], lay["typesize"], dev).decode_from_host([own], out)
            ok = ok and np.array_equal(out.cpu().numpy(), raw[:lay["nbytes"]])
        return ok, (np.dtype(dtype).name, n, kind, fb, bs)

    families = {"deskew": deskew_case, "affine": affine_case, "rl": rl_case, "flatfield": flat_case, "codec": codec_case,
                "blur": blur_case, "estimators": estimator_case, "rl_ysep": rl_ysep_case, "host_twins": host_twin_case,
                "rl_stats": rl_stats_case, "rl_long_z": rl_long_z_case, "rl_fft": rl_fft_case}
    if args.only:
        families = {k: v for k, v in families.items() if k in args.only.split(",")}
    if args.large:
        families = {"rl_large": rl_large_case, "deskew_large": deskew_large_case}
    d.set_spectrum_cache_bytes(0)
    stats = {k: {"cases": 0, "failures": []} for k in families}
    t_end = time.time() + args.seconds
    i = 0
    names = list(families)
    while time.time() < t_end:
        name = names[i % len(names)]
        seed = int(rng.integers(0, 2**31))
        res = families[name](np.random.default_rng(seed))
        i += 1
        if res is None:
            continue
        ok, desc = res
        stats[name]["cases"] += 1
        if not ok and len(stats[name]["failures"]) < 10:
            stats[name]["failures"].append({"seed": seed, "case": repr(desc)})
        if i % 60 == 0:
            print(json.dumps({"progress": {k: v["cases"] for k, v in stats.items()}}), flush=True)
    bad = 0
    for k, v in stats.items():
        bad += len(v["failures"])
        print(json.dumps({"family": k, **v}), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
