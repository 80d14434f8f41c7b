"""What does the first unit of a run pay that the others do not?  (bench config 4: first unit 172 ms of kernels, then 33.)"""
import sys, time, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
import bench
from shrimpy_amd.pipeline import VolumeReconstructor
dev = torch.device("cuda:0")
shape = (2048, 256, 2048)
raw = bench.synthetic_raw(shape, seed=1, device=dev).to(torch.uint16)
torch.cuda.synchronize()
# as in bench: the resident leg has run in this process before the store leg (kernels loaded), then empty_cache()
rec0 = VolumeReconstructor(shape, bench.plate_settings("config4"), dev)
rec0(raw); torch.cuda.synchronize()
del rec0
torch.cuda.empty_cache()
out = {}
t = time.perf_counter(); rec = VolumeReconstructor(shape, bench.plate_settings("config4"), dev); torch.cuda.synchronize(); out["construct_ms"] = round((time.perf_counter() - t) * 1e3, 1)
for k in range(3):
    t = time.perf_counter(); r = rec(raw); out[f"call{k}_host_ms"] = round((time.perf_counter() - t) * 1e3, 1); torch.cuda.synchronize(); out[f"call{k}_total_ms"] = round((time.perf_counter() - t) * 1e3, 1)
    del r
st = torch.cuda.memory_stats(dev)
out["hipMalloc_calls"] = st.get("num_device_alloc"); out["reserved_GB"] = round(st["reserved_bytes.all.current"] / 1e9, 2)
# a bare hipMalloc of 2 GB through the caching allocator, for scale
torch.cuda.empty_cache()
t = time.perf_counter(); x = torch.empty(2_000_000_000, dtype=torch.uint8, device=dev); torch.cuda.synchronize(); out["empty_2GB_fresh_ms"] = round((time.perf_counter() - t) * 1e3, 1)
t = time.perf_counter(); x.zero_(); torch.cuda.synchronize(); out["first_touch_zero_2GB_ms"] = round((time.perf_counter() - t) * 1e3, 1)
t = time.perf_counter(); x.zero_(); torch.cuda.synchronize(); out["second_zero_2GB_ms"] = round((time.perf_counter() - t) * 1e3, 1)
print(json.dumps(out))
