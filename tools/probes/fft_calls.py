"""One call each of the contiguous batched 1-D transforms, for a kernel trace (rocprofv3 --kernel-trace --stats)."""
import torch
dev = torch.device("cuda:0")
Z, Y, X = 180, 2048, 2304
x = torch.rand((Z, Y, X), device=dev)
torch.cuda.synchronize()
f = torch.fft.rfft(x, dim=2); torch.cuda.synchronize()
g = f.transpose(1, 2).contiguous(); torch.cuda.synchronize()
torch.zeros(1, device=dev).add_(1); torch.cuda.synchronize()      # marker: add_
a = torch.fft.fft(g, dim=2); torch.cuda.synchronize()
torch.zeros(1, device=dev).add_(2); torch.cuda.synchronize()
b = torch.fft.ifft(g, dim=2); torch.cuda.synchronize()
torch.zeros(1, device=dev).add_(3); torch.cuda.synchronize()
c = torch.fft.ifft(g, dim=2, norm="forward"); torch.cuda.synchronize()
torch.zeros(1, device=dev).add_(4); torch.cuda.synchronize()
d = torch.fft.irfft(f, n=X, dim=2); torch.cuda.synchronize()
torch.zeros(1, device=dev).add_(5); torch.cuda.synchronize()
e = torch.fft.irfft(f, n=X, dim=2, norm="forward"); torch.cuda.synchronize()
