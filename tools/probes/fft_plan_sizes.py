"""Work-area bytes the axis-by-axis plans of the PCC grid ask for (borrowed from the torch allocator per call),
and what the cached plans themselves hold on the device."""
import ctypes
import json
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from shrimpy_amd import fft3

dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
lib = fft3._lib_hipfft()
Z, Y, X = 180, 2048, 2304
XC = X // 2 + 1
free0 = torch.cuda.mem_get_info()[0]
out = {}
for name, kind, n, batch in (("r2c_x", fft3._HIPFFT_R2C, X, Z * Y), ("c2c_y", fft3._HIPFFT_C2C, Y, Z * XC),
                             ("c2c_z", fft3._HIPFFT_C2C, Z, XC * Y), ("c2r_x", fft3._HIPFFT_C2R, X, Z * Y)):
    plan, work_bytes = fft3._plan(dev, kind, n, batch)
    out[name] = {"work_bytes": work_bytes}
out["device_bytes_held_by_the_four_plans"] = free0 - torch.cuda.mem_get_info()[0]
print(json.dumps(out))
