import sys; sys.path.insert(0, '.')
import torch
from efficientat_amd import ops
dev = torch.device('cuda:0'); torch.manual_seed(0)
B, C, F, T, k, s, act = 256, 672, 8, 63, 5, 2, 2
x = torch.randn(B, C, F, T, device=dev); w = torch.randn(C, k * k, device=dev); b = torch.zeros(C, device=dev)
for _ in range(3): ops.dw_conv(x, w, b, k, s, act, None)
torch.cuda.synchronize()
