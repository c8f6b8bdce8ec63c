import torch
p = torch.nn.Parameter(torch.randn(10, device='cuda')); p.grad = torch.randn(10, device='cuda')
for kw in ({'fused': True}, {'fused': True, 'capturable': True}, {}):
    opt = torch.optim.Adam([p], lr=1e-3, **kw)
    v0 = p._version; opt.step(); print(kw, 'version', v0, '->', p._version)
