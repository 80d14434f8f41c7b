"""Time of lsr_cross_correlate_z_c64 alone on the tracker's grid (LSR_ZCORR_ORDER picks the radix order)."""
import json
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from shrimpy_amd import _lib, fft3

dev = torch.device("cuda:0")
Z, Y, XC = 180, 2048, 1153
f1 = torch.randn((XC, Y, Z), dtype=torch.complex64, device=dev)
g = torch.randn((Z, XC, Y), dtype=torch.complex64, device=dev)
tw = fft3._twiddle_table(Z, dev)


def run():
    _lib.call("lsr_cross_correlate_z_c64", f1.data_ptr(), g.data_ptr(), tw.data_ptr(), Z, Y, XC, _lib.stream_ptr(dev))


run(); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    run()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 10
print(json.dumps({"order": os.environ.get("LSR_ZCORR_ORDER", "default 4,5,3,3"), "ms": ms, "GBps": 3 * 8 * Z * Y * XC / ms / 1e6}))
