"""A/B of the stride-2 depthwise data-gradient kernels at B=256: EAT_DWP_DGRAD2=0/1 python tools/dwd_ab.py  (GPU diagnostic)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientat_amd import ops
dev = torch.device("cuda:0")
for B, C, F, T, k in [(256, 64, 64, 500, 3), (256, 72, 32, 250, 5), (256, 240, 16, 125, 3), (256, 672, 8, 63, 5), (128, 1344, 8, 63, 5)]:
    Fo, To = ops.conv_out(F, k, 2), ops.conv_out(T, k, 2)
    dz = torch.randn(B, C, Fo, To, device=dev); w = torch.randn(C, k * k, device=dev) * 0.2
    for _ in range(3): dx = ops.dw_conv_dgrad(dz, w, (B, C, F, T), k, 2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): dx = ops.dw_conv_dgrad(dz, w, (B, C, F, T), k, 2)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    print(f"B={B} C={C} {F}x{T} k{k}s2: {t*1e3:7.1f} us  {(dx.numel()+dz.numel())*4/1e9/t:5.2f} TB/s", flush=True)
