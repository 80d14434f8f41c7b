"""Where do rocFFT's 24 ms for one rfftn of the PCC grid go?  Times the three axes separately, a few
layouts and shapes (torch.fft = rocFFT)."""
import json

import torch

dev = torch.device("cuda:0")


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


shape = (180, 2048, 2304)
x = torch.rand(shape, device=dev)
res = {}
res["rfftn"] = timed(lambda: torch.fft.rfftn(x))
res["rfft_x"] = timed(lambda: torch.fft.rfft(x, dim=2))
c = torch.fft.rfft(x, dim=2)
res["fft_y"] = timed(lambda: torch.fft.fft(c, dim=1))
res["fft_z"] = timed(lambda: torch.fft.fft(c, dim=0))
res["fft_yz (fftn dims 0,1)"] = timed(lambda: torch.fft.fftn(c, dim=(0, 1)))
res["irfftn"] = timed(lambda: torch.fft.irfftn(torch.fft.rfftn(x), s=shape)) - res["rfftn"]
del c
for alt in [(192, 2048, 2304), (256, 2048, 2304), (180, 2048, 2048), (128, 2048, 2048), (171, 2048, 2270)]:
    try:
        y = torch.rand(alt, device=dev)
        res[f"rfftn {alt}"] = timed(lambda: torch.fft.rfftn(y), 3)
        del y
    except Exception as exc:  # noqa: BLE001
        res[f"rfftn {alt}"] = str(exc)[:80]
print(json.dumps(res, indent=1))
