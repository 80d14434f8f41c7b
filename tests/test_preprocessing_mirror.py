"""CPU tests of the preprocessor adapter; they read like the reference's
``shrimpy/tests/test_preprocessing.py`` (heavy steps monkeypatched, control flow exercised).

The last test runs the REFERENCE's own ``shrimpy.preprocessing`` with this package standing in for
``biahub`` (skipped where /root/reference is absent, e.g. on the GPU box): the drop-in claim,
checked against the caller's real code.
"""

import sys
import types

from pathlib import Path

import numpy as np
import pytest

from shrimpy_amd import preprocessing as pp
from shrimpy_amd.preprocessing import _LabelfreePreprocessor, _settings_kwargs, build_preprocessor

ZYX = (16, 64, 64)
DESKEW = dict(ls_angle_deg=30.0, pixel_size_um=0.1133, scan_step_um=0.15, keep_overhang=False,
              average_n_slices=3)


def test_no_pipeline_returns_none():
    assert build_preprocessor(ZYX, None) is None
    assert build_preprocessor(ZYX, []) is None
    assert build_preprocessor(ZYX, ["sum_projection", "segmentation"]) is None


def test_out_of_scope_steps_fail_loudly():
    with pytest.raises(NotImplementedError, match="phase"):
        build_preprocessor(ZYX, ["deskew", "phase"], deskew=DESKEW)
    with pytest.raises(NotImplementedError):
        build_preprocessor(ZYX, ["vs"])


def test_recon_steps_match_reference(golden_dir):
    ref = np.load(golden_dir / "ref_preprocessing.npz")
    assert tuple(ref["recon_steps"]) == pp.RECON_STEPS


def test_settings_kwargs_filters_to_signature():
    class FakeSettings:
        def model_dump(self):
            return {"a": 1, "b": 2, "unused": 3}

    def func(a, b):
        return a, b

    assert _settings_kwargs(func, FakeSettings()) == {"a": 1, "b": 2}


def _bare(**kw):
    d = dict(zyx_shape=ZYX, deskew_settings=None, output_channel="BF")
    d.update(kw)
    pre = _LabelfreePreprocessor(**d)
    pre._device = None  # keep tensors on CPU: no kernel is called in these tests
    return pre


def test_call_keys_output_channel_and_dtype():
    import torch

    out = _bare()(np.zeros(ZYX, dtype="uint16"))
    assert set(out) == {"BF"} and isinstance(out["BF"], torch.Tensor)
    assert out["BF"].dtype == torch.float32 and tuple(out["BF"].shape) == ZYX


def test_call_runs_flatfield_then_deskew_and_exposes_intermediate(monkeypatch):
    import torch

    from shrimpy_amd.settings import DeskewSettings

    order = []
    pre = _bare(deskew_settings=DeskewSettings(**DESKEW), apply_flatfield=True)
    # with a deskew behind it the flat-field step computes the pattern; the deskew kernel divides
    monkeypatch.setattr(pre, "_flat_field_pattern", lambda v: (order.append("flatfield"), "pattern")[1])
    monkeypatch.setattr(pre, "_deskew", lambda v: (order.append("deskew"), pre._pending_flat_field, v[:4])[2])
    out = pre(np.ones(ZYX, dtype="float32"), label="A/1/0", return_intermediates=True)
    assert order == ["flatfield", "deskew"]
    assert set(out) == {"BF", "deskew"} and out["deskew"] is out["BF"]
    assert tuple(out["BF"].shape) == (4, 64, 64)
    out = pre(np.ones(ZYX, dtype="float32"))
    assert set(out) == {"BF"}
    assert isinstance(out["BF"], torch.Tensor)


def test_step_logs_and_reraises(caplog):
    pre = _bare()

    def boom(_):
        raise ValueError("bad stack")

    with caplog.at_level("ERROR"), pytest.raises(ValueError, match="bad stack"):
        pre._step("[p0] ", "deskew", boom, None)
    assert "[p0] deskew FAILED: bad stack" in caplog.text


def test_flat_field_on_a_cpu_tensor_equals_the_reference_fixture(golden_dir):
    """`_flat_field_BF` of the reference itself (imported in the build container, `oracle/make_golden.py`) on its
    own test input (`shrimpy/tests/test_preprocessing.py:155-160`) and on an odd-Z volume: the host twin's output."""
    import torch

    from shrimpy_amd.flatfield import flat_field_bf, flat_field_pattern

    g = np.load(golden_dir / "ref_preprocessing.npz")
    for k in ("", "_odd"):
        vol = g["flatfield_in" + k]
        out = _bare()._flat_field_BF(torch.as_tensor(vol))
        assert out.device.type == "cpu" and out.dtype == torch.float32
        np.testing.assert_allclose(out.numpy(), g["flatfield_out" + k], rtol=1e-6)
        if np.issubdtype(vol.dtype, np.integer) and vol.min() >= 0 and vol.max() < 65536:   # camera counts go in unconverted
            np.testing.assert_allclose(flat_field_bf(torch.as_tensor(vol.astype(np.uint16))).numpy(), g["flatfield_out" + k],
                                       rtol=1e-6)
    nan = np.ones((4, 2, 3), np.float32)
    nan[2, 1, 1] = np.nan
    pat = flat_field_pattern(torch.as_tensor(nan)).pattern.numpy()
    assert np.isnan(pat[1, 1]) and np.isfinite(np.delete(pat.ravel(), 4)).all()     # quantile propagates NaN


@pytest.mark.gpu
def test_flatfield_matches_reference_capture(golden_dir):
    """The HIP flat-field against what the reference's own ``_flat_field_BF`` produced (captured
    by importing /root/reference/shrimpy/preprocessing.py, ``oracle/make_golden.py``)."""
    import torch

    g = np.load(golden_dir / "ref_preprocessing.npz")
    for k in ("", "_odd"):
        out = _bare()._flat_field_BF(torch.as_tensor(g["flatfield_in" + k], device="cuda"))
        np.testing.assert_allclose(out.cpu().numpy(), g["flatfield_out" + k], rtol=1e-6)


def test_without_a_gpu_the_stages_run_their_host_twins(monkeypatch):
    """The reference's device rule (``shrimpy/preprocessing.py:78-82``): ``cpu`` when no GPU is visible.  The
    stages then run their native host twins (the deskew bit-equal to the oracle); ``require_gpu`` raises."""
    import torch

    from oracle import cpu_ref as o

    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)
    shape = (96, 16, 24)      # scan long enough for the no-overhang window
    pre = build_preprocessor(shape, ["deskew"], deskew=DESKEW, output_channel="BF")
    raw = np.random.default_rng(5).integers(80, 600, shape).astype(np.uint16)
    out = pre(raw, return_intermediates=True)
    assert set(out) == {"BF", "deskew"} and out["BF"].device.type == "cpu" and out["BF"].dtype == torch.float32
    want = o.deskew(raw.astype(np.float32), DESKEW["ls_angle_deg"], round(DESKEW["pixel_size_um"] / DESKEW["scan_step_um"], 3),
                    DESKEW["keep_overhang"], DESKEW["average_n_slices"])
    np.testing.assert_array_equal(out["BF"].numpy(), want)
    with pytest.raises(RuntimeError, match="no HIP device visible"):
        build_preprocessor(shape, ["deskew"], deskew=DESKEW, require_gpu=True)
    # flat-field then deskew, the reference's step order (`shrimpy/preprocessing.py:320-327`), on the host twins
    both = build_preprocessor(shape, ["flatfield", "deskew"], deskew=DESKEW, output_channel="BF")(raw)["BF"]
    corrected = o.flat_field_bf(raw.astype(np.float32))
    want = o.deskew(corrected, DESKEW["ls_angle_deg"], round(DESKEW["pixel_size_um"] / DESKEW["scan_step_um"], 3),
                    DESKEW["keep_overhang"], DESKEW["average_n_slices"])
    np.testing.assert_allclose(both.numpy(), want, rtol=2e-6)


def test_warm_up_resolves_deskewed_shape(monkeypatch):
    import torch

    pre = build_preprocessor.__globals__["_LabelfreePreprocessor"](
        zyx_shape=(2048, 512, 2048), deskew_settings=__import__("shrimpy_amd.settings", fromlist=["x"]).DeskewSettings(**DESKEW),
        output_channel="BF")
    monkeypatch.setattr(pp, "_resolve_device", lambda: torch.device("cuda"))
    pre.warm_up()
    assert pre._zyx_shape == (171, 2048, 2270)


REFERENCE = Path("/root/reference")


@pytest.mark.skipif(not (REFERENCE / "shrimpy" / "preprocessing.py").exists(),
                    reason="the reference is only mounted in the build container")
def test_dropin_through_the_reference_preprocessor(monkeypatch):
    """The reference's ``build_preprocessor`` / ``_LabelfreePreprocessor`` run unmodified with this
    package bound as ``biahub``: settings validation, kwargs filtering by signature, warm-up shape
    call and the deskew call all go through our modules -- and, since round 3, the arithmetic too: the CPU
    tensor the reference hands over runs the native host twin (a spy records the call and passes it on)."""
    import torch

    import shrimpy_amd.deskew as our_deskew
    import shrimpy_amd.settings as our_settings

    from oracle import cpu_ref as o

    biahub = types.ModuleType("biahub")
    biahub.deskew = our_deskew
    biahub.settings = our_settings
    monkeypatch.setitem(sys.modules, "biahub", biahub)
    monkeypatch.setitem(sys.modules, "biahub.deskew", our_deskew)
    monkeypatch.setitem(sys.modules, "biahub.settings", our_settings)
    monkeypatch.syspath_prepend(str(REFERENCE))
    monkeypatch.setattr(sys, "dont_write_bytecode", True)
    ref_pp = __import__("shrimpy.preprocessing", fromlist=["build_preprocessor"])

    calls = {}

    real_launch = our_deskew.deskew_with_matrix

    def spy(raw, matrix, pre_shape, avg=1, out=None, border="constant", **kw):
        calls["args"] = (tuple(raw.shape), pre_shape, avg)
        calls["border"] = border
        res = real_launch(raw, matrix, pre_shape, avg, out=out, border=border, **kw)
        assert isinstance(res, torch.Tensor) and res.device.type == "cpu"
        return res

    monkeypatch.setattr(our_deskew, "deskew_with_matrix", spy)
    raw_shape = (48, 12, 20)
    pre = ref_pp.build_preprocessor(raw_shape, ["deskew"], deskew=dict(DESKEW), output_channel="BF")
    expect_shape, _ = our_deskew.get_deskewed_data_shape(raw_shape, 30.0, 0.755, False, 3)
    assert tuple(pre._zyx_shape) == expect_shape  # the reference's warm_up called OUR shape rule
    raw = np.random.default_rng(0).integers(80, 600, raw_shape).astype(np.uint16)
    out = pre(raw, label="A/1/0", return_intermediates=True)
    assert set(out) == {"BF", "deskew"}
    assert calls["args"] == (raw_shape, (12, 20, expect_shape[2]), 3)
    np.testing.assert_array_equal(out["BF"].numpy(), o.deskew(raw.astype(np.float32), 30.0, 0.755, False, 3))

    # the convention switches ride through the reference's signature filter like any other field
    pre = ref_pp.build_preprocessor(raw_shape, ["deskew"], output_channel="BF",
                                    deskew=dict(DESKEW, orientation="flip_z+rot90", border="grid-constant"))
    want = o.deskew(raw.astype(np.float32), 30.0, 0.755, False, 3, orientation="flip_z+rot90",
                    border="grid-constant")
    assert tuple(pre._zyx_shape) == want.shape       # warm_up saw the rotated shape
    out = pre(raw)
    assert calls["border"] == "grid-constant"
    np.testing.assert_array_equal(out["BF"].numpy(), want)
