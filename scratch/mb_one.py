import sys; sys.path.insert(0, '.')
import torch
from efficientat_amd import ops
dev = torch.device('cuda:0'); torch.manual_seed(0)
B = 256
Ci, Ce, Co, F_, T, k, s, act = 16, 64, 24, 64, 500, 3, 2, 1
x = torch.randn(B, Ci, F_, T, device=dev)
we = ops.pw_prepack(torch.randn(Ce, Ci, device=dev) / Ci ** 0.5, None); be = torch.zeros(Ce, device=dev)
wd = torch.randn(Ce, k * k, device=dev) / k; bd = torch.zeros(Ce, device=dev)
wp = ops.pw_prepack(torch.randn(Co, Ce, device=dev) / Ce ** 0.5, None); bp = torch.zeros(Co, device=dev)
for _ in range(3):
    ops.mbconv(x, we, be, wd, bd, wp, bp, Ce, Co, k, s, act)
torch.cuda.synchronize()
