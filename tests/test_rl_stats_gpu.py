"""Richardson-Lucy reduction scalars (VERDICT r3 row g; north-star "wavefront reductions for the ratio / normalisation"):
``flux = sum x_new * H^T 1``, ``change = sum |x_new - x|``, ``total = sum x_new`` per iteration, summed by the kernels in
the epilogue that writes the new estimate.  Checked here on every RL path against the oracle's estimates summed in fp64
(``oracle.rl_iteration_scalars``), relative 1e-5 -- the f32 partial sums of a thread are good to ~1e-6 and independent,
the cross-workgroup sums are fp64.  No reference code for RL (``/root/reference/docs/data_structure.md:58-62``)."""

import numpy as np
import pytest

from oracle import cpu_ref as o

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def _t(a, device):
    import torch

    return torch.as_tensor(np.ascontiguousarray(a), device=device)


def _check(stats, want, iterations, rtol=RTOL):
    assert stats.iterations == iterations
    for name in ("flux", "change", "total"):
        got, ref = getattr(stats, name), want[name][:iterations]
        assert got.shape == (iterations,)
        np.testing.assert_allclose(got, ref, rtol=rtol, err_msg=name)


PATHS = [
    # (id, psf builder, plan kwargs, expected plan.path)
    ("fused", lambda: o.gaussian_psf((9, 7, 7), (2.0, 1.2, 1.2))[0], dict(), "fused"),
    ("separable", lambda: o.gaussian_psf((9, 7, 7), (2.0, 1.2, 1.2))[0], dict(fused="never"), "separable"),
    ("y-separable", lambda: o.rotated_psf((9, 7, 7), (2.0, 1.2, 1.2), 30.0), dict(fused="never"), "y-separable"),
    ("y-separable-fused", lambda: o.rotated_psf((9, 7, 7), (2.0, 1.2, 1.2), 30.0), dict(), "y-separable (fused)"),
    ("y-separable-4", lambda: o.rotated_psf((7, 11, 5), (1.6, 2.0, 1.0), 25.0), dict(), "y-separable (4 launches)"),
    ("dense", lambda: o.rotated_psf((9, 7, 7), (2.0, 1.2, 1.2), 30.0), dict(separable="never"), "dense"),
    ("generic", lambda: o.rotated_psf((13, 5, 5), (2.5, 1.0, 1.0), 30.0), dict(separable="never"), "generic"),
]


@pytest.mark.parametrize("name,make_psf,kw,path", PATHS, ids=[p[0] for p in PATHS])
@pytest.mark.parametrize("vshape", [(20, 44, 150), (5, 33, 64)])
def test_rl_scalars_match_the_oracle_on_every_path(device, name, make_psf, kw, path, vshape):
    import torch

    from shrimpy_amd.deconvolve import RichardsonLucyPlan

    psf = make_psf()
    y = o.bead_scene(vshape, seed=4100 + vshape[0], psf=o.gaussian_psf((5, 5, 5), (1.2, 1.0, 1.0))[0], density=1e-3)
    iters = 5
    want = o.rl_iteration_scalars(y, psf, iters)
    plan = RichardsonLucyPlan(vshape, psf, device, **kw)
    if path is not None:
        assert plan.path == path
    elif not plan.fused_ysep:
        pytest.skip("no one-launch ky (x) kzx specialisation for this PSF")
    yd = _t(y, device)
    plain = plan(yd, iterations=iters)
    assert plan.last_stats is None
    x = plan(yd, iterations=iters, stats=True)
    assert torch.equal(x, plain), "asking for the scalars must not change the estimate"
    _check(plan.last_stats, want, iters)
    assert tuple(plan.stats_device.shape) == (iters, 3) and plan.stats_device.dtype == torch.float64
    # flux conservation: sum x_new * H^T 1 = sum y * Hx / (Hx + eps) -> sum y
    np.testing.assert_allclose(plan.last_stats.flux, float(y.astype(np.float64).sum()), rtol=2e-6)
    # a padded y, an explicit x0 and an even / odd number of iterations (the working volumes ping-pong)
    x0 = _t(np.full(vshape, float(y.mean()), np.float32), device)
    x = plan(yd, iterations=2, x0=x0, stats=True)
    _check(plan.last_stats, o.rl_iteration_scalars(y, psf, 2, x0=np.full(vshape, float(y.mean()), np.float32)), 2)


@pytest.mark.parametrize("fused", ["always", "never"])
def test_rl_scalars_equal_fp64_reductions_of_the_estimates_for_every_compiled_extent(device, fused):
    """Every template instance of the one-launch and the two-launch kernels (tap counts 3 .. 15 per axis, all tile
    classes), awkward volume extents: the epilogue's sums against torch's fp64 reductions of consecutive estimates.
    (Round 4 found the sums of ONE instance corrupted by prefetches still in flight at the kernel's end -- the
    reduction's temporaries had been given the registers those loads land in; see RlStats::pin.)"""
    import torch

    from shrimpy_amd import _lib
    from shrimpy_amd.deconvolve import RichardsonLucyPlan

    rng = np.random.default_rng(11)
    odd = [3, 5, 7, 9, 11, 13, 15]
    cases = [(a, b, b) for a in odd for b in odd]
    for i, pshape in enumerate(cases):
        if fused == "always" and not _lib.call_value("lsr_rl_sep_fused_supported", *pshape):
            continue
        vshape = (int(rng.integers(6, 30)), int(rng.choice([17, 40, 64, 70])), int(rng.choice([64, 100, 128, 200])))
        factors = [np.abs(rng.normal(1.0, 0.4, n)).astype(np.float32) + 0.05 for n in pshape]
        factors = [f / f.sum() for f in factors]
        y = _t((rng.random(vshape) * 80 + 1).astype(np.float32), device)
        plan = RichardsonLucyPlan(vshape, None, device, psf_factors=factors, fused=fused)
        xs = [y] + [plan(y, iterations=n) for n in (1, 2, 3)]
        plan(y, iterations=3, stats=True)
        s = plan.last_stats
        for n in range(3):
            tot = float(xs[n + 1].double().sum())
            chg = float((xs[n + 1].double() - xs[n].double()).abs().sum())
            assert abs(s.total[n] / tot - 1) < RTOL and abs(s.change[n] / chg - 1) < RTOL, (pshape, vshape, n, s)
        np.testing.assert_allclose(s.flux, float(y.double().sum()), rtol=RTOL, err_msg=str((pshape, vshape)))


def test_rl_tol_stops_early_and_returns_that_iterations_estimate(device):
    import torch

    from shrimpy_amd.deconvolve import RichardsonLucyPlan, richardson_lucy

    psf, factors = o.gaussian_psf((9, 7, 7), (2.0, 1.2, 1.2))
    vshape = (16, 40, 130)
    y = o.bead_scene(vshape, seed=77, psf=psf, density=1e-3)
    want = o.rl_iteration_scalars(y, psf, 30)
    rel = want["change"] / want["total"]
    tol = float(0.5 * (rel[7] + rel[8]))            # first met by iteration index 8
    first = int(np.argmax(rel <= tol))
    assert first == 8
    for kw in (dict(), dict(fused="never")):
        plan = RichardsonLucyPlan(vshape, None, device, psf_factors=factors, **kw)
        x = plan(_t(y, device), iterations=30, tol=tol)
        s = plan.last_stats
        # the criterion is read one iteration behind the launches: at most one more ran
        assert s.stopped_by_tol and s.iterations in (first + 1, first + 2), s.iterations
        assert s.rel_change[first] <= tol < s.rel_change[first - 1]
        ref = plan(_t(y, device), iterations=s.iterations)
        assert torch.equal(x, ref), "the estimate returned is the one of the last iteration that ran"
        _check(s, want, s.iterations)
    # tol never met: all iterations run; tol = 0 is never met by a changing estimate
    x, s = richardson_lucy(_t(y, device), psf, iterations=4, tol=0.0, return_stats=True)
    assert s.iterations == 4 and not s.stopped_by_tol
    assert torch.equal(x, richardson_lucy(_t(y, device), psf, iterations=4))
    with pytest.raises(ValueError):
        richardson_lucy(_t(y, device), psf, iterations=4, tol=-1.0)


def test_rl_scalars_of_an_all_zero_stack_are_zero(device):
    """The all-zero-stack case of the reference's integration test (shrimpy/tests/test_mantis_integration.py:285-341) must
    still return zeros -- and zero scalars; tol then stops at once (total == 0 counts as converged)."""
    import torch

    from shrimpy_amd.deconvolve import richardson_lucy

    psf, _ = o.gaussian_psf((9, 7, 7), (2.0, 1.2, 1.2))
    y = torch.zeros((6, 20, 70), device=device)
    x, s = richardson_lucy(y, psf, iterations=5, return_stats=True)
    assert float(x.abs().max()) == 0.0 and torch.isfinite(x).all()
    assert not s.flux.any() and not s.change.any() and not s.total.any()
    x, s = richardson_lucy(y, psf, iterations=5, tol=1e-3, return_stats=True)
    assert s.stopped_by_tol and s.iterations <= 2 and float(x.abs().max()) == 0.0


def test_iterate_padded_hands_the_scalars_to_the_slab_split(device):
    """The one-iteration building block of shrimpy_amd.slab carries the scalars of its own volume."""
    import torch

    from shrimpy_amd.deconvolve import PaddedVolume, RichardsonLucyPlan

    psf, factors = o.gaussian_psf((5, 5, 5), (1.2, 1.0, 1.0))
    vshape = (12, 40, 100)
    y = o.bead_scene(vshape, seed=9, psf=psf, density=2e-3)
    plan = RichardsonLucyPlan(vshape, None, device, psf_factors=factors)
    assert plan.fused
    ypad, a, b = plan.new_padded_input(), plan.new_padded_input(), plan.new_padded_input()
    ypad.view.copy_(_t(y, device))
    a.view.copy_(ypad.view)
    acc = torch.full((3,), 5.0, dtype=torch.float64, device=device)
    plan.iterate_padded(ypad, a, b, stats=acc)
    want = o.rl_iteration_scalars(y, psf, 1)
    # the RL loop entries own (zero) their scalars; one iteration through iterate_padded is such a loop of length 1,
    # so the 5.0 is gone
    np.testing.assert_allclose(acc.cpu().numpy(), [want["flux"][0], want["change"][0], want["total"][0]], rtol=RTOL)
    with pytest.raises(ValueError):
        plan.iterate_padded(ypad, a, b, stats=torch.zeros(3, device=device))   # float32: refused
    assert isinstance(ypad, PaddedVolume)


def test_tol_means_the_same_on_cpu_and_gpu_tensors(device):
    """``richardson_lucy(y, tol=...)`` stopped one iteration earlier on a CPU tensor than on a GPU tensor.  Both now
    return the estimate one iteration past the first that met ``tol`` (the device reads iteration i's scalars while i + 1
    runs): the same iteration count and, within the RL bar, the same estimate, for the separable and the dense route."""
    import torch

    from shrimpy_amd.deconvolve import richardson_lucy

    psf, _ = o.gaussian_psf((5, 5, 7), (1.1, 0.9, 1.4))
    rot = o.rotated_psf((5, 5, 7), (1.1, 0.9, 1.4), 30.0)
    y = o.bead_scene((12, 20, 26), seed=2, psf=psf, density=2e-3)
    for kernel in (psf, rot):
        want = o.rl_iteration_scalars(y, kernel, 8)
        rel = want["change"] / want["total"]
        for first in (2, 4):
            tol = float(0.5 * (rel[first - 1] + rel[first]))
            xc, sc = richardson_lucy(torch.as_tensor(y), kernel, iterations=8, tol=tol, return_stats=True)
            xg, sg = richardson_lucy(_t(y, device), kernel, iterations=8, tol=tol, return_stats=True)
            assert sc.iterations == sg.iterations == first + 2 and sc.stopped_by_tol and sg.stopped_by_tol
            a, b = xc.numpy().astype(np.float64), xg.cpu().numpy().astype(np.float64)
            assert np.all(np.abs(a - b) <= 2e-4 * np.abs(a) + 1e-4 * np.abs(a).max())
