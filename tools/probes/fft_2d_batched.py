"""Would a batched 2-D transform over (Z, Y) in layout [XC][Z][Y] beat C2C(Y) + transpose + C2C(Z)?
Kernel-level view: rocprofv3 --kernel-trace on this script."""
import torch
dev = torch.device("cuda:0")
Z, Y, XC = 180, 2048, 1153
c = torch.randn((XC, Z, Y), dtype=torch.complex64, device=dev)
torch.cuda.synchronize()
for _ in range(2):
    torch.zeros(1, device=dev).add_(1); torch.cuda.synchronize()
    a = torch.fft.fftn(c, dim=(1, 2)); torch.cuda.synchronize()
    del a
torch.zeros(1, device=dev).add_(2); torch.cuda.synchronize()
