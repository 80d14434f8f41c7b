import torch,time
d=torch.device('cuda')
x=torch.empty((171,2048,2270),device=d).normal_()
y=torch.empty_like(x)
for name,fn,bytes_ in [('copy',lambda: y.copy_(x), 2*x.numel()*4), ('add3',lambda: torch.add(x,y,out=y), 3*x.numel()*4)]:
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/10
    print(name, 'ms',ms,'GB/s',bytes_/ms/1e6)
