"""A/B of the depthwise forward kernels on the large training planes at B=256: EAT_DWP_TILE=0/1 python tools/dw_ab2.py  (GPU diagnostic)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientat_amd import ops
dev = torch.device("cuda:0")
B = 256
for C, F, T, k, s in [(16, 64, 500, 3, 1), (64, 64, 500, 3, 2), (72, 32, 250, 3, 1), (72, 32, 250, 5, 2)]:
    x = torch.randn(B, C, F, T, device=dev); w = torch.randn(C, k * k, device=dev) * 0.2; b = torch.zeros(C, device=dev)
    for _ in range(3): y = ops.dw_conv(x, w, b, k, s, ops.ACT_NONE)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): y = ops.dw_conv(x, w, b, k, s, ops.ACT_NONE)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    print(f"C={C} {F}x{T} k{k}s{s}: {t*1e3:7.1f} us  {(x.numel()+y.numel())*4/1e9/t:5.2f} TB/s", flush=True)
