"""Does the lane decoder slow down when something else uses the card?  (In the streamed run a decode takes 21-28 ms,
alone 6.9 ms: tools/probes/gpu_timeline.py.)

    python tools/probes/decode_interference.py

Times lsr_blosc_decode_device on the config-4 camera stack (HIP events on its stream) alone, beside a pinned host ->
device copy, beside a device -> pinned host copy, beside both, and beside fused RL launches on another stream.
One JSON line.
"""

from __future__ import annotations

import json
import sys
import threading
import time

from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))


def main():
    import torch

    import bench
    from shrimpy_amd.io import codecs
    from shrimpy_amd.io.device_codec import DeviceBloscDecoder

    dev = torch.device("cuda:0")
    shape = (2048, 256, 2048)
    raw = bench.synthetic_raw(shape, seed=4000, device=dev).to(torch.uint16).cpu().numpy()
    zc = 32
    fb = zc * shape[1] * shape[2] * 2
    with ThreadPoolExecutor(16) as pool:
        frames = list(pool.map(lambda i: codecs.blosc_encode(raw[i:i + zc], 2, "zstd", 1, codecs.SHUFFLE_BYTE, 32768, backend="lsrecon"),
                               range(0, shape[0], zc)))
    dec = DeviceBloscDecoder(raw.nbytes, fb, 32768, 2, dev)
    table = np.zeros((len(frames), 2), np.int64)
    at = 0
    for f, fr in enumerate(frames):
        table[f] = (at, len(fr))
        at += len(fr)
    comp = torch.frombuffer(bytearray(b"".join(frames)), dtype=torch.uint8).to(dev)
    tab = torch.as_tensor(table).to(dev)
    out = torch.empty(shape, dtype=torch.uint16, device=dev)
    dec.decode(comp, at, tab, out)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), raw)

    side = torch.cuda.Stream(dev)
    side2 = torch.cuda.Stream(dev)
    host_a = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
    host_b = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
    dev_a = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    dev_b = torch.empty(1 << 30, dtype=torch.uint8, device=dev)

    def timed_decode(reps=3):
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dec.decode(comp, at, tab, out)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        return round(float(np.median(ts)), 2)

    res = {"alone_ms": timed_decode()}

    def beside(label, enqueue):
        stop = threading.Event()

        def feeder():
            while not stop.is_set():
                enqueue()
                side.synchronize()
                side2.synchronize()
        th = threading.Thread(target=feeder)
        th.start()
        time.sleep(0.05)
        res[label] = timed_decode()
        stop.set()
        th.join()
        torch.cuda.synchronize()

    def h2d():
        with torch.cuda.stream(side):
            dev_a.copy_(host_a, non_blocking=True)

    def d2h():
        with torch.cuda.stream(side2):
            host_b.copy_(dev_b, non_blocking=True)

    beside("beside_h2d_copy_ms", h2d)
    beside("beside_d2h_copy_ms", d2h)
    beside("beside_both_copies_ms", lambda: (h2d(), d2h()))

    # the stager's own slots: hipHostMalloc of exactly the size (staging._pinned_tensor), 1.32 GB, offset views
    from shrimpy_amd.staging import _pinned_tensor

    n = 1_322_000_000
    host_c = _pinned_tensor((n + 4096,), torch.uint8)
    dev_c = torch.empty(n + 4096, dtype=torch.uint8, device=dev)

    def d2h_exact():
        with torch.cuda.stream(side2):
            host_c[:n].copy_(dev_c[:n], non_blocking=True)

    def h2d_exact():
        with torch.cuda.stream(side):
            dev_c[:n].copy_(host_c[:n], non_blocking=True)
    beside("beside_d2h_into_hipHostMalloc_slot_ms", d2h_exact)
    beside("beside_h2d_from_hipHostMalloc_slot_ms", h2d_exact)

    # fused RL launches on another stream
    from shrimpy_amd.pipeline import VolumeReconstructor

    rec = VolumeReconstructor(shape, bench.plate_settings("config4"), dev)
    stack = out.clone()

    def rl():
        with torch.cuda.stream(side):
            rec(stack)
    beside("beside_deskew_and_rl_ms", rl)
    res["alone_again_ms"] = timed_decode()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
