"""Phase cross-correlation on the config-2 deskewed grid, ms per call (reference spectrum cached and not):
    python3 tools/pcc_time.py [--grid 171,2048,2270]      (LSR_RFFT_ROWS=8 forces the 8-row tiles of the x legs)"""
import argparse
import json
import os
import sys

from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", default="171,2048,2270")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import torch

    from shrimpy_amd import dynatrack as d

    dev = torch.device("cuda:0")
    shape = tuple(int(v) for v in args.grid.split(","))
    g = torch.Generator(device=dev).manual_seed(9)
    vol = torch.rand(shape, device=dev, generator=g) * 900 + 100
    mov = torch.roll(vol, (2, -5, 7), dims=(0, 1, 2)).contiguous()

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.reps):
            out = fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / args.reps, out

    d.set_spectrum_cache_bytes(0)
    ms, shift = timed(lambda: d._phase_cross_corr(vol, mov))
    d.set_spectrum_cache_bytes(8 << 30)
    ms_cached, shift2 = timed(lambda: d._phase_cross_corr(vol, mov))
    print(json.dumps({"grid": list(shape), "rows_env": os.environ.get("LSR_RFFT_ROWS", "auto"), "ms": ms, "ms_reference_cached": ms_cached,
                      "found": list(shift), "found_cached": list(shift2), "rolled_by": [2, -5, 7]}))


if __name__ == "__main__":
    main()
