"""A/B of the depthwise kernels on the mn10 late-layer geometries at B=256: EAT_DWP=0/1 python tools/dw_ab.py  (GPU diagnostic)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientat_amd import ops
dev = torch.device("cuda:0")
B = 256
cases = [(200, 8, 63, 3, 1), (672, 8, 63, 3, 1), (120, 16, 125, 5, 1), (672, 8, 63, 5, 2), (240, 16, 125, 3, 2), (960, 4, 32, 5, 1)]
for C, F, T, k, s in cases:
    x = torch.randn(B, C, F, T, device=dev)
    w = torch.randn(C, k * k, device=dev) * 0.2
    b = torch.randn(C, device=dev) * 0.1
    pool = torch.zeros(B, C, device=dev)
    for _ in range(3): y = ops.dw_conv(x, w, b, k, s, ops.ACT_HSWISH, pool)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): y = ops.dw_conv(x, w, b, k, s, ops.ACT_HSWISH, pool)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20
    gb = (x.numel() + y.numel()) * 4 / 1e9
    print(f"C={C} {F}x{T} k{k}s{s}: {t*1e3:7.1f} us  {gb/t*1e3/1e3:5.2f} TB/s", flush=True)
