"""Phase cycle stamps of decode_blocks_kernel (probe build of csrc/blosc_decode.hip with -DLSR_DEC_PROBE, via LSR_LIBRARY)."""
import ctypes, sys
from pathlib import Path
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
import bench
from shrimpy_amd import _lib
from shrimpy_amd.io import codecs
from shrimpy_amd.io.device_codec import DeviceBloscDecoder
dev = torch.device("cuda:0")
shape = (512, 256, 2048)
raw = bench.synthetic_raw(shape, seed=4000, device=dev).to(torch.uint16).cpu().numpy()
zc = 32; fb = zc * shape[1] * shape[2] * 2
with ThreadPoolExecutor(16) as pool:
    frames = list(pool.map(lambda i: codecs.blosc_encode(raw[i:i + zc], 2, "zstd", 1, codecs.SHUFFLE_BYTE, 32768, backend="lsrecon"), range(0, shape[0], zc)))
dec = DeviceBloscDecoder(raw.nbytes, fb, 32768, 2, dev)
out = torch.empty(shape, dtype=torch.uint16, device=dev)
dec.decode_from_host(frames, out); dec.decode_from_host(frames, out)
assert np.array_equal(out.cpu().numpy(), raw)
lib = _lib.load()
buf = np.zeros(1024 * 8, np.int64)
lib.lsr_debug_dec_probe.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.lsr_debug_dec_probe(buf.ctypes.data, buf.size) == 0
t = buf.reshape(1024, 8)[:256]
d = np.diff(t[:, :7], axis=1)
names = ["literals header + Huffman table", "sequence header + tables", "sequences -> workspace", "seek + literals (4 streams)", "matches", "tail (sizes, flags)"]
print("median cycles per phase over", len(t), "waves (lane 0); total", np.median(t[:, 6] - t[:, 0]))
print("match loop iterations / nseq of lane 0:", np.median(t[:, 7] // 1000000), np.median(t[:, 7] % 1000000), "max iters", (t[:, 7] // 1000000).max())
for n, col in zip(names, d.T):
    print(f"  {n:34s} median {np.median(col):10.0f}  p90 {np.percentile(col, 90):10.0f}")
