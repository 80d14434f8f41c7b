"""The chunk codecs that run on the device (round-4 verdict item 1).

Encoder (``csrc/blosc_encode.hip``): the frames are a subset of zstd -- Huffman-coded literals, RLE and raw blocks --
inside c-blosc 1.x frames, the format the acquisition engine writes (``shrimpy/mantis/mantis_engine.py:474-481``).

* CPU tests: the host twin (``lsr_blosc_encode_device_cpu``, the same code as the kernel's serial steps plus loops
  for its parallel ones) against two independent decoders -- the system libzstd behind ``lsr_blosc_decode_host`` and the
  pure-Python frame walker -- on distributions that exercise every branch: RLE planes, raw planes, the 11-bit length
  limit, tree descriptions as nibbles and as an FSE stream, ragged last frames and blocks.
* GPU tests: the kernels write the SAME BYTES as the twin, on the same inputs and on a hot-path result.
"""

import ctypes

import numpy as np
import pytest

from shrimpy_amd import _lib
from shrimpy_amd.io import codecs
from shrimpy_amd.io.device_codec import encode_frames_host, plan_frames


def _rl_like(n, seed=3):
    """float32 with the byte statistics of a deconvolved volume: few exponents, busy mantissas."""
    rng = np.random.default_rng(seed)
    return (100.0 + 30.0 * rng.standard_normal(n) ** 2 + (rng.random(n) < 1e-3) * rng.uniform(200, 4000, n)).astype(np.float32)


def _fibonacci_bytes():
    counts = [1, 1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 233, 377, 610, 987, 1597, 2584, 4181, 6765, 10946, 17711]
    arr = np.concatenate([np.full(c, i, np.uint8) for i, c in enumerate(counts)])
    np.random.default_rng(0).shuffle(arr)
    return arr[:arr.size // 4 * 4].copy()


def _cases():
    rng = np.random.default_rng(1)
    p_half = 0.5 ** np.arange(1, 40)
    p_half /= p_half.sum()
    return [
        ("rl-like f32, 1 MB frames", _rl_like(700_000), 1 << 20, 0),
        ("rl-like f32, ragged frames, 64 K blocks", _rl_like(300_001), 700_004, 65536),
        ("camera counts u16", rng.poisson(110, 600_000).astype(np.uint16), 1 << 20, 0),
        ("camera counts u16, 32 K blocks", rng.poisson(110, 600_000).astype(np.uint16), 1 << 19, 32768),
        ("noise u8 (raw planes, verbatim blocks)", rng.integers(0, 256, 1_000_001, dtype=np.uint8), 1 << 19, 0),
        ("zeros (RLE planes)", np.zeros(1 << 18, np.float32), 1 << 19, 0),
        ("signed f32 (256 symbols in the top plane)", (rng.standard_normal(1 << 18) * 1e3).astype(np.float32), 1 << 19, 0),
        ("3 symbols", rng.integers(0, 3, 1 << 19, dtype=np.uint8), 1 << 18, 0),
        ("geometric u8 (FSE-coded weights)", rng.geometric(0.02, 1 << 19).astype(np.uint8), 1 << 18, 0),
        ("2^-k distribution (length limit)", rng.choice(39, 1 << 19, p=p_half).astype(np.uint8), 1 << 18, 0),
        ("exact Fibonacci counts (deepest tree)", _fibonacci_bytes(), 46364, 0),
        ("128 even symbols", (rng.integers(0, 256, 1 << 19, dtype=np.uint8) // 2 * 2).astype(np.uint8), 1 << 18, 0),
        ("200 symbols, uniform", rng.integers(0, 200, 1 << 19).astype(np.uint8), 1 << 18, 0),
    ] + [(f"{n} bytes", rng.geometric(0.1, n // 4).astype(np.float32), n, 0)
         for n in (4, 8, 2048 * 4 - 4, 2048 * 4, 2048 * 4 + 4, 100_000, 262_144 + 4)]


CASES = _cases()


def _want(raw, f, frame_bytes):
    out = np.zeros(frame_bytes, np.uint8)
    seg = raw[f * frame_bytes:(f + 1) * frame_bytes]
    out[:seg.size] = seg
    return out


@pytest.mark.parametrize("label,arr,frame_bytes,blocksize", CASES, ids=[c[0] for c in CASES])
def test_host_twin_frames_decode_with_libzstd_and_python(label, arr, frame_bytes, blocksize):
    raw = arr.reshape(-1).view(np.uint8)
    frames = encode_frames_host(arr, frame_bytes, blocksize)
    assert len(frames) == -(-raw.size // frame_bytes)
    for f, frame in enumerate(frames):
        h = codecs.blosc_header(frame)
        assert (h["nbytes"], h["cbytes"], h["typesize"], h["compressor"]) == (frame_bytes, len(frame), arr.dtype.itemsize, "zstd")
        assert h["flags"] & 0x10 and bool(h["flags"] & 0x1) == (arr.dtype.itemsize > 1)
        want = _want(raw, f, frame_bytes)
        assert np.array_equal(codecs.blosc_decode(frame, backend="lsrecon"), want), "system libzstd disagrees"
        got = np.empty(frame_bytes, np.uint8)
        codecs._py_blosc_decode(frame, got)
        assert np.array_equal(got, want), "the Python walker disagrees"


def test_host_twin_matches_zstd_level_1_on_image_bytes():
    """The subset (order-0 literals only) is within a few percent of what zstd level 1 finds on shuffled float32."""
    x = _rl_like(1 << 20)
    ours = sum(len(f) for f in encode_frames_host(x, 1 << 22, 0))
    zstd1 = len(codecs.blosc_encode(x, 4, "zstd", 1, codecs.SHUFFLE_BYTE, 256 * 1024))
    assert ours < 1.03 * zstd1, (ours, zstd1)
    assert ours < 0.85 * x.nbytes


def test_plan_rejects_bad_arguments():
    with pytest.raises(_lib.LsrError):
        plan_frames(1000, 3, 1000)                # typesize
    with pytest.raises(_lib.LsrError):
        plan_frames(1001, 2, 1000)                # not whole elements
    with pytest.raises(_lib.LsrError):
        plan_frames(1 << 20, 4, 1 << 20, 1 << 20)  # more than 64 K elements per plane
    n, scratch, cap = plan_frames(10 << 20, 4, 3 << 20)
    assert n == 4 and cap >= n * (3 << 20) and scratch > 10 << 20


@pytest.mark.gpu
@pytest.mark.parametrize("label,arr,frame_bytes,blocksize", CASES, ids=[c[0] for c in CASES])
def test_device_frames_equal_the_host_twin(device, label, arr, frame_bytes, blocksize):
    import torch

    from shrimpy_amd.io.device_codec import DeviceBloscEncoder

    raw = arr.reshape(-1).view(np.uint8)
    want = encode_frames_host(arr, frame_bytes, blocksize)
    enc = DeviceBloscEncoder(raw.size, arr.dtype.itemsize, frame_bytes, device, blocksize)
    t = torch.as_tensor(arr.reshape(-1).view({1: np.uint8, 2: np.int16, 4: np.float32}[arr.dtype.itemsize])).to(device)
    got = enc.encode_to_host(t)
    assert len(got) == len(want)
    for f, (g, w) in enumerate(zip(got, want)):
        assert g == w, f"frame {f}: device bytes differ from the twin (sizes {len(g)} / {len(w)})"
    # and a second call on the same encoder (buffers reused)
    assert enc.encode_to_host(t) == want


@pytest.mark.gpu
def test_device_encoder_on_a_hot_path_result(device):
    """deskew -> RL result of a bead scene, encoded on the device, decoded by libzstd: the volume back bit for bit."""
    import torch

    from oracle import cpu_ref as o
    from shrimpy_amd.deconvolve import richardson_lucy
    from shrimpy_amd.deskew import fast_deskew_zyx
    from shrimpy_amd.io.device_codec import DeviceBloscEncoder

    psf, _ = o.gaussian_psf((9, 7, 7), (2.0, 1.2, 1.2))
    raw = o.bead_scene((384, 64, 256), seed=4000, psf=psf)
    d = fast_deskew_zyx(raw_data=torch.as_tensor(raw, device=device), ls_angle_deg=30.0, px_to_scan_ratio=0.755,
                        keep_overhang=False, average_n_slices=3)
    x = richardson_lucy(d, psf, iterations=20).contiguous()
    zc = 4
    frame_bytes = zc * x.shape[1] * x.shape[2] * 4
    enc = DeviceBloscEncoder(x.numel() * 4, 4, frame_bytes, device)
    frames = enc.encode_to_host(x)
    host = x.cpu().numpy()
    total = 0
    for f, frame in enumerate(frames):
        got = codecs.blosc_decode(frame, backend="lsrecon").view(np.float32)
        want = np.zeros(frame_bytes // 4, np.float32)
        seg = host[f * zc:(f + 1) * zc].reshape(-1)
        want[:seg.size] = seg
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        total += len(frame)
    assert total < 0.8 * host.nbytes
