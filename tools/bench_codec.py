"""Time the device-side chunk codecs on a volume of a BASELINE configuration's size.

    python tools/bench_codec.py [--shape 86 2048 2491] [--zc 3] [--reps 5] [--check]

``encode``: a float32 volume with the byte statistics of a Richardson-Lucy result (or, with ``--from-pipeline``, the
result of deskew + 20 RL iterations of a synthetic config-4 stack) -> blosc-zstd frames of ``zc`` planes each
(``lsr_blosc_encode_device``).  Prints one JSON line: milliseconds per volume (HIP events on the launch stream), GB/s of
source bytes, the compression ratio next to host zstd level 1 on a sample, and the per-kernel split.
"""

from __future__ import annotations

import argparse
import json
import sys
import time

from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def rl_like(shape, device, seed=0):
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    n = int(np.prod(shape))
    x = torch.randn(n, generator=g, device=device)
    x = 100.0 + 30.0 * x * x
    beads = torch.rand(n, generator=g, device=device) < 1e-3
    x = torch.where(beads, x + 2000.0 * torch.rand(n, generator=g, device=device), x)
    return x.reshape(shape).contiguous()


def bench_decode(args, dev):
    """A config-4 camera stack (2048, 256, 2048) uint16 as 64 chunks of 32 planes, each a blosc-zstd frame written by the
    HOST zstd encoder at level 1 with 32 KB blocks (what c-blosc, and so the acquisition, writes) -> decoded on the device."""
    import torch

    from concurrent.futures import ThreadPoolExecutor

    import bench

    from shrimpy_amd import _lib
    from shrimpy_amd.io import codecs
    from shrimpy_amd.io.device_codec import DeviceBloscDecoder

    shape = (2048, 256, 2048)
    raw = bench.synthetic_raw(shape, seed=4000, device=dev).to(torch.uint16).cpu().numpy()
    zc = 32
    fb = zc * shape[1] * shape[2] * 2
    t0 = time.perf_counter()
    with ThreadPoolExecutor(16) as pool:
        frames = list(pool.map(lambda i: codecs.blosc_encode(raw[i:i + zc], 2, "zstd", 1, codecs.SHUFFLE_BYTE, args.decode_blocksize,
                                                             backend="lsrecon"), range(0, shape[0], zc)))
    t_enc = time.perf_counter() - t0
    t0 = time.perf_counter()
    with ThreadPoolExecutor(16) as pool:
        outs = list(pool.map(lambda f: codecs.blosc_decode(f, backend="lsrecon"), frames))
    t_host = time.perf_counter() - t0
    del outs
    dec = DeviceBloscDecoder(raw.nbytes, fb, args.decode_blocksize, 2, dev)
    table = np.zeros((len(frames), 2), np.int64)
    at = 0
    for f, fr in enumerate(frames):
        table[f] = (at, len(fr))
        at += (len(fr) + 15) // 16 * 16
    blob = np.zeros(at, np.uint8)
    for (o, n), fr in zip(table, frames):
        blob[o:o + n] = np.frombuffer(fr, np.uint8)
    comp = torch.as_tensor(blob).to(dev)
    tab = torch.as_tensor(table).to(dev)
    out = torch.empty(shape, dtype=torch.uint16, device=dev)
    dec.decode(comp, at, tab, out)
    torch.cuda.synchronize()
    dec.check(int(dec.status.cpu().item()))
    assert np.array_equal(out.cpu().numpy(), raw), "device decode differs from the stack"
    stream = torch.cuda.current_stream(dev)
    times = []
    for _ in range(args.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        dec.decode(comp, at, tab, out)
        e1.record(stream)
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    print(json.dumps({
        "what": "lsr_blosc_decode_device", "shape": list(shape), "dtype": "uint16", "frames": len(frames), "frame_bytes": fb,
        "blocksize": args.decode_blocksize, "blocks": len(frames) * (fb // args.decode_blocksize),
        "compressed_bytes": int(table[:, 1].sum()), "ratio": round(float(table[:, 1].sum()) / raw.nbytes, 4),
        "ms": round(float(np.median(times)), 3), "ms_all": [round(t, 3) for t in times],
        "decoded_GBps": round(raw.nbytes / (np.median(times) * 1e-3) / 1e9, 1),
        "host_16_threads_decode_s": round(t_host, 3), "host_16_threads_encode_s": round(t_enc, 3),
        "checked": "bit-equal to the stack", "library": _lib.library_source_sha16()}))


def main():
    import torch

    from shrimpy_amd import _lib
    from shrimpy_amd.io import codecs
    from shrimpy_amd.io.device_codec import DeviceBloscEncoder

    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", type=int, nargs=3, default=[86, 2048, 2491])
    ap.add_argument("--zc", type=int, default=3, help="planes per chunk (frame)")
    ap.add_argument("--blocksize", type=int, default=0)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--check", action="store_true", help="decode every frame with libzstd and compare")
    ap.add_argument("--from-pipeline", action="store_true", help="encode a real deskew + RL result (config-4 unit)")
    ap.add_argument("--decode", action="store_true", help="time lsr_blosc_decode_device on a config-4 camera stack instead")
    ap.add_argument("--decode-blocksize", type=int, default=32768, help="blosc block size of the frames (c-blosc's own for zstd level 1)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    if args.decode:
        return bench_decode(args, dev)
    if args.from_pipeline:
        sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
        import bench

        from shrimpy_amd.pipeline import VolumeReconstructor

        raw = bench.synthetic_raw((2048, 256, 2048), seed=4000, device=dev).to(torch.uint16)
        x = VolumeReconstructor((2048, 256, 2048), bench.plate_settings("config4"), dev)(raw).contiguous()
        del raw
    else:
        x = rl_like(tuple(args.shape), dev)
    shape = tuple(x.shape)
    frame_bytes = args.zc * shape[1] * shape[2] * 4
    enc = DeviceBloscEncoder(x.numel() * 4, 4, frame_bytes, dev, args.blocksize)
    enc.encode(x)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream(dev)
    times = []
    for _ in range(args.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        out, frames = enc.encode(x)
        e1.record(stream)
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    table = frames.cpu().numpy()
    total = int(table[:, 1].sum())
    line = {
        "what": "lsr_blosc_encode_device", "shape": list(shape), "source": "pipeline result" if args.from_pipeline else "rl-like synthetic",
        "frames": int(enc.n_frames), "frame_bytes": frame_bytes, "blocksize": args.blocksize or 4 * 65536,
        "ms": round(float(np.median(times)), 3), "ms_all": [round(t, 3) for t in times],
        "source_GBps": round(x.numel() * 4 / (np.median(times) * 1e-3) / 1e9, 1),
        "compressed_bytes": total, "ratio": round(total / (x.numel() * 4), 4),
        "library": _lib.library_source_sha16(),
    }
    # host zstd level 1 on the first frame's bytes: the ratio the CLI's host encoder gets
    sample = x.reshape(-1)[: frame_bytes // 4].cpu().numpy()
    t0 = time.perf_counter()
    host = codecs.blosc_encode(sample, 4, "zstd", 1, codecs.SHUFFLE_BYTE, 256 * 1024)
    line["host_zstd1_ratio_first_frame"] = round(len(host) / sample.nbytes, 4)
    line["host_zstd1_s_per_GB_one_thread"] = round((time.perf_counter() - t0) / (sample.nbytes / 1e9), 3)
    line["device_ratio_first_frame"] = round(int(table[0, 1]) / frame_bytes, 4)
    if args.check:
        end = int(table[-1, 0] + table[-1, 1])
        hostbuf = out[:end].cpu().numpy()
        ref = x.cpu().numpy().reshape(-1)
        for f, (o, n) in enumerate(table):
            got = codecs.blosc_decode(hostbuf[o:o + n].tobytes(), backend="lsrecon").view(np.float32)
            want = np.zeros(frame_bytes // 4, np.float32)
            seg = ref[f * (frame_bytes // 4):(f + 1) * (frame_bytes // 4)]
            want[:seg.size] = seg
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"frame {f} does not decode to the volume"
        line["checked"] = "every frame decoded by libzstd equals the volume"
    print(json.dumps(line))


if __name__ == "__main__":
    main()
