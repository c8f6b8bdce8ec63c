import torch
dev='cuda:0'
def t(f,n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
for mb in (100, 400, 1600):
    n = mb*1024*1024//4
    a = torch.empty(n, device=dev); b = torch.randn(n, device=dev)
    us = t(lambda: a.fill_(1.0)); print(f'fill {mb} MB: {us:.1f} us {mb*1.048576e6/us*1e-6:.2f} TB/s write')
    us = t(lambda: a.copy_(b)); print(f'copy {mb} MB: {us:.1f} us {2*mb*1.048576e6/us*1e-6:.2f} TB/s r+w')
    us = t(lambda: b.sum()); print(f'sum  {mb} MB: {us:.1f} us {mb*1.048576e6/us*1e-6:.2f} TB/s read')
