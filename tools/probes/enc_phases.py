"""Phase cycle stamps of encode_blocks_kernel (probe build: csrc/blosc_encode.hip with -DLSR_ENC_PROBE, loaded through LSR_LIBRARY)."""
import ctypes, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from shrimpy_amd import _lib
from shrimpy_amd.io.device_codec import DeviceBloscEncoder
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from bench_codec import rl_like
dev = torch.device("cuda:0")
x = rl_like((86, 2048, 2491), dev)
enc = DeviceBloscEncoder(x.numel() * 4, 4, 3 * 2048 * 2491 * 4, dev)
enc.encode(x); torch.cuda.synchronize(); enc.encode(x); torch.cuda.synchronize()
lib = _lib.load()
out = np.zeros(1024 * 16, np.int64)
lib.lsr_debug_enc_probe.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.lsr_debug_enc_probe(out.ctypes.data, out.size) == 0
t = out.reshape(1024, 16)
d = np.diff(t[:, :11], axis=1)
names = ["histogram", "rank+init", "merge", "depth..codes", "description", "plane0", "plane1", "plane2", "plane3", "tail"]
names = ["histogram", "rank sort", "merge(lane0)", "depths,lengths,codes", "description(lane0)", "->plane0 done", "plane1", "plane2", "plane3", "end"]
print("median cycles (s_memtime, 100 MHz?) per phase over 1024 workgroups; total", np.median(t[:, 10] - t[:, 0]))
for n, col in zip(names, d.T):
    print(f"  {n:24s} median {np.median(col):10.0f}  p90 {np.percentile(col, 90):10.0f}")
