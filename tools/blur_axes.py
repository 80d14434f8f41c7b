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
        import ctypes
        ms = timed(lambda: _lib.call("lsr_blur_reflect_f32", x.data_ptr(), y.data_ptr(), *shape, axis, taps.data_ptr(), r, ctypes.c_float(0.0), ctypes.c_float(0.0), torch.cuda.current_stream().cuda_stream))
        print(json.dumps({"r": r, "axis": axis, "ms": ms, "GBps": 8 * x.numel() / ms / 1e6}))
import ctypes
for r in (8, 16):
    taps = torch.ones(2 * r + 1, device=dev) / (2 * r + 1)
    ms = timed(lambda: _lib.call("lsr_blur_reflect_yx_f32", x.data_ptr(), y.data_ptr(), *shape, taps.data_ptr(), r, taps.data_ptr(), r, torch.cuda.current_stream().cuda_stream))
    print(json.dumps({"r": r, "axis": "y+x fused", "ms": ms, "GBps": 8 * x.numel() / ms / 1e6}))
    # bit-equality with the two separate passes
    t1, t2 = torch.empty_like(x), torch.empty_like(x)
    _lib.call("lsr_blur_reflect_f32", x.data_ptr(), t1.data_ptr(), *shape, 1, taps.data_ptr(), r, ctypes.c_float(0.0), ctypes.c_float(0.0), torch.cuda.current_stream().cuda_stream)
    _lib.call("lsr_blur_reflect_f32", t1.data_ptr(), t2.data_ptr(), *shape, 2, taps.data_ptr(), r, ctypes.c_float(0.0), ctypes.c_float(0.0), torch.cuda.current_stream().cuda_stream)
    _lib.call("lsr_blur_reflect_yx_f32", x.data_ptr(), y.data_ptr(), *shape, taps.data_ptr(), r, taps.data_ptr(), r, torch.cuda.current_stream().cuda_stream)
    print(json.dumps({"r": r, "fused == two passes (bit for bit)": bool(torch.equal(t2, y))}))
from shrimpy_amd.dynatrack import _gaussian_blur_3d
for sigma in (2.0, 4.0):
    ms = timed(lambda: _gaussian_blur_3d(x, sigma), reps=5)
    print(json.dumps({"gaussian_blur_3d sigma": sigma, "ms": ms}))
ms = timed(lambda: y.copy_(x))
print(json.dumps({"copy_ms": ms, "GBps": 8 * x.numel() / ms / 1e6}))
