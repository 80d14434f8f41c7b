"""rocFFT spends most of a 3-D real transform of the PCC grid in transposes and pre/post passes
(tools/probes/fft_split.py, profiles/r02_secondary_kernel_stats.csv).  Prototype: one contiguous batched 1-D
transform per axis with explicit transposes in between, the spectrum left in the transposed layout
[XC][Y][Z]; forward + inverse against rfftn / irfftn."""
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
Z, Y, X = shape
x = torch.rand(shape, device=dev)


def fwd(v):
    f = torch.fft.rfft(v, dim=2)                       # [Z][Y][XC]
    f = f.transpose(1, 2).contiguous()                 # [Z][XC][Y]
    f = torch.fft.fft(f, dim=2)
    f = f.reshape(Z, -1).t().contiguous().reshape(f.shape[1], Y, Z)   # [XC][Y][Z]
    return torch.fft.fft(f, dim=2)


def inv(s):
    f = torch.fft.ifft(s, dim=2)                       # [XC][Y][Z]
    xc = f.shape[0]
    f = f.reshape(-1, Z).t().contiguous().reshape(Z, xc, Y)          # [Z][XC][Y]
    f = torch.fft.ifft(f, dim=2)
    f = f.transpose(1, 2).contiguous()                 # [Z][Y][XC]
    return torch.fft.irfft(f, n=X, dim=2)


res = {}
ref = torch.fft.rfftn(x)
mine = fwd(x)
res["fwd_max_rel_err"] = float((mine.permute(2, 1, 0) - ref).abs().max() / ref.abs().max())
back = inv(mine)
res["roundtrip_max_err"] = float((back - x).abs().max())
del ref, mine, back
res["rfftn_ms"] = timed(lambda: torch.fft.rfftn(x))
res["axiswise_fwd_ms"] = timed(lambda: fwd(x))
s = fwd(x)
r = torch.fft.rfftn(x)
res["irfftn_ms"] = timed(lambda: torch.fft.irfftn(r, s=shape))
res["axiswise_inv_ms"] = timed(lambda: inv(s))
f = torch.fft.rfft(x, dim=2)
res["rfft_x_ms"] = timed(lambda: torch.fft.rfft(x, dim=2))
res["transpose_last2_ms"] = timed(lambda: f.transpose(1, 2).contiguous())
g = f.transpose(1, 2).contiguous()
res["fft_y_contig_ms"] = timed(lambda: torch.fft.fft(g, dim=2))
res["transpose_outer_ms"] = timed(lambda: g.reshape(Z, -1).t().contiguous())
h = g.reshape(Z, -1).t().contiguous().reshape(g.shape[1], Y, Z)
res["fft_z_contig_ms"] = timed(lambda: torch.fft.fft(h, dim=2))
res["irfft_x_ms"] = timed(lambda: torch.fft.irfft(f, n=X, dim=2))
print(json.dumps(res, indent=1))
