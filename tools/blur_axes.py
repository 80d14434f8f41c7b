import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shrimpy_amd import _lib
dev = torch.device("cuda", 0)
shape = (171, 2048, 2270)
x = torch.rand(shape, device=dev); y = torch.empty_like(x)
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for r in (8, 20):
    taps = torch.ones(2 * r + 1, device=dev) / (2 * r + 1)
    for axis in (0, 1, 2):
        ms = timed(lambda: _lib.call("lsr_blur_reflect_f32", x.data_ptr(), y.data_ptr(), *shape, axis, taps.data_ptr(), r, 0.0, 0.0, torch.cuda.current_stream().cuda_stream))
        print(json.dumps({"r": r, "axis": axis, "ms": ms, "GBps": 8 * x.numel() / ms / 1e6}))
ms = timed(lambda: y.copy_(x))
print(json.dumps({"copy_ms": ms, "GBps": 8 * x.numel() / ms / 1e6}))
